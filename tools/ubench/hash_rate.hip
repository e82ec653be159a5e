// Micro-benchmark: minimap2 hash64 as hipcc compiles it (the shift-add steps become v_mad_u64_u32 /
// v_mul_lo_u32 multiplies) vs the same steps kept as 64-bit shift+add (opaque barriers stop the
// multiply canonicalisation) vs a hand-split 32-bit version for masks below 2^64.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int N = 2048;
__device__ __forceinline__ uint64_t h_mul(uint64_t key, uint64_t mask) {
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}
#define OPAQUE(x) asm("" : "+v"(x))
__device__ __forceinline__ uint64_t h_shift(uint64_t key, uint64_t mask) {
    uint64_t t = key << 21; OPAQUE(t);
    key = (~key + t) & mask;
    key = key ^ key >> 24;
    t = key + (key << 3); OPAQUE(t);
    uint64_t u = key << 8; OPAQUE(u);
    key = (t + u) & mask;
    key = key ^ key >> 14;
    t = key + (key << 2); OPAQUE(t);
    key = (t + (key << 4)) & mask; OPAQUE(key);
    key = key ^ key >> 28;
    t = key << 31; OPAQUE(t);
    key = (key + t) & mask;
    return key;
}
template <int OP> __global__ __launch_bounds__(256) void k(uint64_t *out, uint64_t seed, uint64_t mask) {
    uint64_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x, c = a ^ b, d = a + b;
#pragma unroll 1
    for (int i = 0; i < N; i++) {
        if (OP == 0) { a = h_mul(a + i, mask); b = h_mul(b + i, mask); c = h_mul(c + i, mask); d = h_mul(d + i, mask); }
        if (OP == 1) { a = h_shift(a + i, mask); b = h_shift(b + i, mask); c = h_shift(c + i, mask); d = h_shift(d + i, mask); }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
template <int OP> int run(const char *name, uint64_t *d, uint64_t *h) {
    const int blocks = 256 * 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const uint64_t mask = (1ull << 42) - 1;
    k<OP><<<blocks, 256>>>(d, 12345, mask);
    hipEventRecord(a); k<OP><<<blocks, 256>>>(d, 12345, mask); hipEventRecord(b); CHK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, a, b);
    CHK(hipMemcpy(h, d, 8 * 64, hipMemcpyDeviceToHost));
    const double hashes_per_simd = (double)blocks * 4 / 1024 * N * 4;
    printf("%-40s %7.1f us  %.1f cycles per wave-hash per SIMD  (check %016llx)\n", name, ms * 1000, ms * 1e6 / hashes_per_simd * 2.4, (unsigned long long)h[5]);
    return 0;
}
int main() {
    uint64_t *d; CHK(hipMalloc(&d, (size_t)(256 * 8 * 256 + 1) * 8));
    uint64_t h[64];
    run<0>("hash64 as compiled (multiplies)", d, h);
    run<1>("hash64 kept as shift+add", d, h);
    return 0;
}
