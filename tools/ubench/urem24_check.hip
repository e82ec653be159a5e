// urem24_check — does `x % m` come out right when the compiler can PROVE both operands are below 2^24 and m is a RUN-TIME value?
// hipcc (ROCm 7.2, gfx950) then expands the remainder in fp32: q = trunc(float(x) * v_rcp_iflag_f32(float(m))), one correction
// step for an UNDER-estimated quotient, none for an over-estimated one.  Every x in [0, 2^24) against every m in [1, 64] (m from a
// kernel argument: a compile-time m becomes a multiplication by a magic constant, which is exact), compared with the 64-bit
// remainder of operands whose range the compiler cannot see.  (Found by the LDS-order probe of hulk_countmin.hip, round 6.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long *bad, unsigned *first, unsigned m_arg, unsigned hide) {
    const unsigned x24 = (blockIdx.x * blockDim.x + threadIdx.x) & 0xFFFFFFu;            // provably < 2^24
    const unsigned m24 = 1u + (m_arg & 63u);                                              // provably < 2^24, not a constant
    const unsigned r24 = x24 % m24;
    unsigned long long xo = x24, mo = m24;
    asm volatile("" : "+v"(xo), "+v"(mo));                                                // ranges hidden: the 64-bit expansion
    const unsigned ref = (unsigned)(xo % (mo + hide));
    if (r24 != ref) {
        atomicAdd(bad, 1ull);
        const unsigned i = atomicAdd(&first[0], 1u);
        if (i < 8) { first[1 + 3 * i] = x24; first[2 + 3 * i] = m24; first[3 + 3 * i] = r24; }
    }
}
int main() {
    unsigned long long *d_bad, bad = 0; unsigned *d_first, first[25] = {0};
    (void)hipMalloc((void **)&d_bad, 8); (void)hipMalloc((void **)&d_first, sizeof first);
    (void)hipMemset(d_bad, 0, 8); (void)hipMemset(d_first, 0, sizeof first);
    for (unsigned m = 0; m < 64; m++) hipLaunchKernelGGL(k, dim3((1u << 24) / 256), dim3(256), 0, 0, d_bad, d_first, m, 0u);
    (void)hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(first, d_first, sizeof first, hipMemcpyDeviceToHost);
    printf("urem24_check: x %% m, both operands provably < 2^24, m a run-time value; x in [0, 2^24), m in [1, 64]: %llu of %llu remainders wrong\n", bad, (1ull << 24) * 64);
    for (unsigned i = 0; i < first[0] && i < 8; i++)
        printf("  %u %% %u = %u by hipcc's 24-bit expansion (the remainder is %u)\n", first[1 + 3 * i], first[2 + 3 * i], first[3 + 3 * i], first[1 + 3 * i] % first[2 + 3 * i]);
    return 0;
}
