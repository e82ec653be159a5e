// lds_atomic_order — in what order does the LDS apply the lanes of ONE ds_add_rtn_f64 that hit the same address?
// k_cmsd_freq (hulk_countmin.hip) takes the value an atomic add RETURNS as "the counter before this bin": that is the
// bin-order replay of the count-min sketch only if same-address lanes are applied in ascending lane order and successive
// instructions of a wave in program order.  Patterns: all 64 lanes on one address; pairs; a random partition into groups;
// two instructions back to back on overlapping addresses.  Prints the number of lanes whose returned value is not the sum of
// the lower lanes' (and the earlier instruction's) operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const int *p0, const int *p1, double *r0, double *r1, int rounds) {
    __shared__ double s[64];
    const int l = threadIdx.x;
    for (int r = 0; r < rounds; r++) {
        s[l] = 0.0;
        __syncthreads();
        const double a = atomicAdd(&s[p0[r * 64 + l]], (double)(1 + l));
        const double b = atomicAdd(&s[p1[r * 64 + l]], (double)(1000 + l));
        r0[r * 64 + l] = a; r1[r * 64 + l] = b;
        __syncthreads();
    }
}
int main() {
    const int R = 4096;
    std::vector<int> p0(R * 64), p1(R * 64);
    srand(1);
    for (int r = 0; r < R; r++)
        for (int l = 0; l < 64; l++) {
            const int mode = r % 4;
            p0[r * 64 + l] = mode == 0 ? 0 : mode == 1 ? l / 2 : rand() % (1 + r % 61);
            p1[r * 64 + l] = mode == 0 ? 0 : mode == 1 ? (63 - l) / 2 : rand() % (1 + r % 59);
        }
    int *d0, *d1; double *e0, *e1;
    hipMalloc((void **)&d0, R * 64 * 4); hipMalloc((void **)&d1, R * 64 * 4); hipMalloc((void **)&e0, R * 64 * 8); hipMalloc((void **)&e1, R * 64 * 8);
    hipMemcpy(d0, p0.data(), R * 64 * 4, hipMemcpyHostToDevice); hipMemcpy(d1, p1.data(), R * 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d0, d1, e0, e1, R);
    std::vector<double> r0(R * 64), r1(R * 64);
    hipMemcpy(r0.data(), e0, R * 64 * 8, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), e1, R * 64 * 8, hipMemcpyDeviceToHost);
    long bad0 = 0, bad1 = 0;
    for (int r = 0; r < R; r++) {
        double s[64] = {0};
        for (int l = 0; l < 64; l++) { if (r0[r * 64 + l] != s[p0[r * 64 + l]]) bad0++; s[p0[r * 64 + l]] += 1 + l; }
        for (int l = 0; l < 64; l++) { if (r1[r * 64 + l] != s[p1[r * 64 + l]]) bad1++; s[p1[r * 64 + l]] += 1000 + l; }
    }
    printf("ds_add_rtn_f64, %d rounds of two instructions: lanes out of ascending-lane order: first instruction %ld, second %ld (of %d each)\n", R, bad0, bad1, R * 64);
    return (bad0 || bad1) ? 1 : 0;
}
