// Exhaustive check: how many Newton steps after v_rcp_f64 give RN(1/r) for EVERY integer r in [1, 2^31]?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int STEPS> __global__ void k(unsigned long long *bad, unsigned long long *first) {
    unsigned b = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; r <= 0x80000000ull; r += (uint64_t)gridDim.x * blockDim.x) {
        const double d = (double)(uint32_t)r;
        double y = __builtin_amdgcn_rcp(d);
        for (int s = 0; s < STEPS; s++) { const double e = __builtin_fma(-d, y, 1.0); y = __builtin_fma(y, e, y); }
        if (y != 1.0 / d) { b++; atomicMin(first, (unsigned long long)r); }
    }
    if (b) atomicAdd(bad, (unsigned long long)b);
}
template <int STEPS> void run() {
    unsigned long long *d, h[2] = {0, ~0ull};
    hipMalloc(&d, 16); hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
    k<STEPS><<<4096, 256>>>(d, d + 1); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("rcp + %d Newton step(s): %llu mismatches of 2^31 (first r = %llu)\n", STEPS, h[0], h[1]);
    hipFree(d);
}
int main() { run<0>(); run<1>(); run<2>(); return 0; }
