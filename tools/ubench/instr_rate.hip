// Micro-benchmark: issue cost (cycles per wave64 instruction) of the VALU ops the hot kernels lean on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int N = 4096;
template <int OP> __global__ __launch_bounds__(256) void k(uint64_t *out, uint64_t seed) {
    uint64_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x, c = a ^ b, d = a + b;   // 4 independent chains
    double fa = (double)(a & 0xffff) + 1.5, fb = (double)(b & 0xffff) + 2.5, fc = fa + 1, fd = fb + 1;
    uint32_t x = (uint32_t)a, y = (uint32_t)b, z = (uint32_t)c, w = (uint32_t)d;
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < N; i++) {
        if (OP == 0) { x = x * 5u + 1u; y = y * 5u + 1u; z = z * 5u + 1u; w = w * 5u + 1u; x ^= y; z ^= w; }                  // mul_lo / mad_u32
        if (OP == 1) { a = a * 2862933555777941757ull + 1; b = b * 2862933555777941757ull + 1; c = c * 2862933555777941757ull + 1; d = d * 2862933555777941757ull + 1; }
        if (OP == 2) { fa = __builtin_fma(fa, 1.0000001, 0.5); fb = __builtin_fma(fb, 1.0000001, 0.5); fc = __builtin_fma(fc, 1.0000001, 0.5); fd = __builtin_fma(fd, 1.0000001, 0.5); }
        if (OP == 3) { fa = __builtin_amdgcn_rcp(fa) + 1.5; fb = __builtin_amdgcn_rcp(fb) + 1.5; fc = __builtin_amdgcn_rcp(fc) + 1.5; fd = __builtin_amdgcn_rcp(fd) + 1.5; }
        if (OP == 4) { x = (x << 3) ^ y; y = (y >> 5) + z; z = (z & w) | x; w = w + x; }                                  // plain 32-bit int ops
        if (OP == 5) { a = (a << 21) + b; b = (b >> 24) ^ c; c = (c << 3) + d; d = (d >> 14) ^ a; }                          // 64-bit shifts/adds
        if (OP == 6) { fa = (double)(uint32_t)x + fa; x = (uint32_t)(int32_t)fb; fb = __builtin_trunc(fb * 1.5) + 1.0; fb = fb > 1e9 ? 2.5 : fb; }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + x + y + z + w + (uint64_t)(fa + fb + fc + fd);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * blockDim.x] = (uint64_t)(t1 - t0);
}
template <int OP> int run(const char *name, int ops_per_iter, uint64_t *d) {
    const int blocks = 256 * 8;          // 8 waves per SIMD: throughput, not latency
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<OP><<<blocks, 256>>>(d, 12345);
    hipEventRecord(a); k<OP><<<blocks, 256>>>(d, 12345); hipEventRecord(b); CHK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, a, b);
    // waves = blocks*4; per SIMD = waves/1024; instr per wave = N*ops
    const double instr_per_simd = (double)blocks * 4 / 1024 * N * ops_per_iter;
    printf("%-44s %7.1f us   %.2f ns per wave-instr per SIMD  (= %.1f cycles @2.4GHz)\n", name, ms * 1000, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    return 0;
}
int main() {
    uint64_t *d; CHK(hipMalloc(&d, (size_t)(256 * 8 * 256 + 1) * 8));
    run<4>("32-bit shift/xor/add (8 ops)", 8, d);
    run<0>("32-bit mul+add x4, xor x2 (6 'ops')", 6, d);
    run<1>("64-bit key*C+1 x4 (4 'ops')", 4, d);
    run<5>("64-bit shift+add/xor x4 (8 ops)", 8, d);
    run<2>("fp64 fma x4", 4, d);
    run<3>("fp64 rcp + add x4 (8 ops)", 8, d);
    run<6>("cvt u32->f64, f64->i32, trunc, mul, add, cmp (8)", 8, d);
    return 0;
}
