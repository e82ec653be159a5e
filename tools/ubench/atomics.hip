// Micro-benchmark: throughput of random atomicAdd into a k^4-bin histogram under different scopes /
// per-XCD private copies.  Build: hipcc --offload-arch=gfx950 -O3 atomics.hip -o atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }
__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *hist, int B, int per_thread, uint32_t *xcc_census) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t *h = hist;
    const uint32_t x = xcc_id();
    if (MODE >= 2) h = hist + (size_t)x * B;                 // private copy per XCD
    if (threadIdx.x == 0) atomicAdd(&xcc_census[x], 1u);
    for (int i = 0; i < per_thread; i++) {
        const uint32_t bin = (uint32_t)(mix(gid * 1000003ull + i) % (uint64_t)B);
        if (MODE == 0 || MODE == 2) atomicAdd(&h[bin], 1u);                                        // agent scope
        else if (MODE == 1 || MODE == 3) __hip_atomic_fetch_add(&h[bin], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 4) __hip_atomic_fetch_add(&h[bin], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        else if (MODE == 5) __hip_atomic_fetch_add(&h[bin], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void k_sum(const uint32_t *hist, int B, int copies, unsigned long long *out) {
    unsigned long long s = 0;
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)B * copies; i += (size_t)gridDim.x * blockDim.x) s += hist[i];
    atomicAdd(out, s);
}
template <int MODE> int run(const char *name, uint32_t *d_hist, int B, uint32_t *d_census, unsigned long long *d_out) {
    const int blocks = 4096, per = 26;                         // 27.3M atomics ~ 1M reads
    CHK(hipMemset(d_hist, 0, (size_t)B * 8 * 4)); CHK(hipMemset(d_census, 0, 64)); CHK(hipMemset(d_out, 0, 8));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(d_hist, B, per, d_census);      // warm
    CHK(hipMemset(d_hist, 0, (size_t)B * 8 * 4)); CHK(hipMemset(d_census, 0, 64));
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) k<MODE><<<blocks, 256>>>(d_hist, B, per, d_census);
    hipEventRecord(b); CHK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, a, b);
    k_sum<<<256, 256>>>(d_hist, B, 8, d_out);
    unsigned long long tot = 0; CHK(hipMemcpy(&tot, d_out, 8, hipMemcpyDeviceToHost));
    uint32_t cen[16]; CHK(hipMemcpy(cen, d_census, 64, hipMemcpyDeviceToHost));
    const double n = 5.0 * blocks * 256 * per;
    printf("%-34s %8.1f us per 27.3M atomics  %6.1f G/s  sum %s (%llu / %.0f)  xcc census:", name, ms * 1000 / 5, n / ms / 1e6,
           (double)tot == n ? "OK" : "MISMATCH", tot, n);
    for (int i = 0; i < 8; i++) printf(" %u", cen[i]);
    printf("\n");
    return 0;
}
int main() {
    const int B = 194481;
    uint32_t *d_hist, *d_census; unsigned long long *d_out;
    CHK(hipMalloc(&d_hist, (size_t)B * 8 * 4)); CHK(hipMalloc(&d_census, 64)); CHK(hipMalloc(&d_out, 8));
    run<0>("agent scope, shared hist", d_hist, B, d_census, d_out);
    run<1>("workgroup scope, shared hist", d_hist, B, d_census, d_out);
    run<2>("agent scope, per-XCD hist", d_hist, B, d_census, d_out);
    run<3>("workgroup scope, per-XCD hist", d_hist, B, d_census, d_out);
    run<4>("wavefront scope, shared hist", d_hist, B, d_census, d_out);
    run<5>("system scope, shared hist", d_hist, B, d_census, d_out);
    return 0;
}
