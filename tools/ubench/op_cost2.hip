// Second, independent measurement of the VALU issue cost on gfx950 (VERDICT r02 "corroborate or retire the 4-cycle
// claim"): op_cost.hip times whole launches with HIP events and divides by a NOMINAL 2.4 GHz.  Here
//   * every wave brackets its own unrolled block with s_memtime (the constant 100 MHz reference counter, read inside the
//     kernel: no launch overhead, no tail), and the median over the waves is taken;
//   * the core clock is MEASURED in the same process by a block whose cycle count is known by construction
//     (s_nop 15 = 16 idle cycles, issued back to back by ONE wave per SIMD) — printed, and to be compared with
//     GRBM_GUI_ACTIVE / wall time of the rocprofv3 pass over this binary (tools/gpu_op_cost.sh);
//   * each op runs at 1 wave per SIMD (latency-exposed unless the 8 register chains cover it) and at 8 waves per SIMD
//     (issue-bound), in its VOP2 (32-bit encoding) and VOP3 (_e64) forms where both exist, plus the packed fp32 ops.
// Output: cycles per wave64 instruction per SIMD = median ticks x 10 ns x measured clock / instructions issued on the SIMD.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int N = 1024;       // loop trips; 8 chains each

__device__ __forceinline__ uint64_t memtime() {
    uint64_t t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

#define OP32(NAME, ASM)                                                                                   \
    __global__ __launch_bounds__(256) void NAME(uint64_t *ticks, uint32_t *out, uint32_t seed) {          \
        uint32_t a[8];                                                                                    \
        for (int i = 0; i < 8; i++) a[i] = seed * (i + 3) + threadIdx.x;                                   \
        uint32_t b = seed | 1u;                                                                           \
        const uint64_t t0 = memtime();                                                                    \
        _Pragma("unroll 1") for (int it = 0; it < N; it++) {                                              \
            _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b));       \
        }                                                                                                 \
        const uint64_t t1 = memtime();                                                                    \
        uint32_t s = 0; for (int i = 0; i < 8; i++) s += a[i];                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                   \
        if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;               \
    }
#define OP64(NAME, ASM)                                                                                   \
    __global__ __launch_bounds__(256) void NAME(uint64_t *ticks, uint32_t *out, uint32_t seed) {          \
        uint64_t a[8];                                                                                    \
        for (int i = 0; i < 8; i++) a[i] = 0x3ff0000000000000ull + ((uint64_t)(seed * (i + 3) + threadIdx.x) << 20);   \
        uint64_t b = 0x3ff0000000000123ull + seed;                                                        \
        const uint64_t t0 = memtime();                                                                    \
        _Pragma("unroll 1") for (int it = 0; it < N; it++) {                                              \
            _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b));       \
        }                                                                                                 \
        const uint64_t t1 = memtime();                                                                    \
        uint64_t s = 0; for (int i = 0; i < 8; i++) s += a[i];                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(s ^ (s >> 32));                           \
        if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;               \
    }

// known cycle count: 8 x s_nop 15 per trip = 128 idle cycles + the loop's own s_add/s_cmp/s_cbranch (counted as 3 issue
// cycles; the printed clock is a lower bound by that much: < 3 %)
__global__ __launch_bounds__(256) void k_nop(uint64_t *ticks, uint32_t *out, uint32_t seed) {
    const uint64_t t0 = memtime();
#pragma unroll 1
    for (int it = 0; it < N; it++) {
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");
    }
    const uint64_t t1 = memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = seed;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

OP32(k_add_u32_e32, "v_add_u32_e32 %0, %0, %1")
OP32(k_add_u32_e64, "v_add_u32_e64 %0, %0, %1")
OP32(k_xor_b32_e32, "v_xor_b32_e32 %0, %0, %1")
OP32(k_xor_b32_e64, "v_xor_b32_e64 %0, %0, %1")
OP32(k_mul_f32_e32, "v_mul_f32_e32 %0, %0, %1")
OP32(k_mul_f32_e64, "v_mul_f32_e64 %0, %0, %1")
OP32(k_add_f32_e32, "v_add_f32_e32 %0, %0, %1")
OP32(k_fmac_f32_e32, "v_fmac_f32_e32 %0, %0, %1")
OP32(k_fma_f32, "v_fma_f32 %0, %0, %1, %1")
OP32(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
OP32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7")
OP32(k_bfe, "v_bfe_u32 %0, %0, 3, 20")
OP32(k_perm, "v_perm_b32 %0, %0, %1, %1")
OP32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
OP32(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %1")
OP32(k_dpp_shr1, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
OP32(k_min_u32, "v_min_u32_e32 %0, %0, %1")
OP32(k_cndmask, "v_cndmask_b32_e32 %0, %0, %1, vcc")
OP64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %1")
OP64(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
OP64(k_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
OP64(k_fma_f64, "v_fma_f64 %0, %0, %1, %1")
OP64(k_mul_f64, "v_mul_f64 %0, %0, %1")
OP64(k_min_f64, "v_min_f64 %0, %0, %1")
OP64(k_rcp_f64, "v_rcp_f64 %0, %0")
OP64(k_trunc_f64, "v_trunc_f64 %0, %0")
OP64(k_lshl_b64, "v_lshlrev_b64 %0, 3, %0")
OP64(k_cmp_u64, "v_cmp_lt_u64 vcc, %0, %1")
#define OPX(NAME, ASM)                                                                                    \
    __global__ __launch_bounds__(256) void NAME(uint64_t *ticks, uint32_t *out, uint32_t seed) {          \
        uint64_t a[8]; uint32_t x[8];                                                                     \
        for (int i = 0; i < 8; i++) { a[i] = 0x3ff0000000000000ull + ((uint64_t)(seed * (i + 3) + threadIdx.x) << 20); x[i] = seed * i + threadIdx.x; } \
        uint32_t b = seed | 1u;                                                                           \
        const uint64_t t0 = memtime();                                                                    \
        _Pragma("unroll 1") for (int it = 0; it < N; it++) {                                              \
            _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(a[i]), "+v"(x[i]) : "v"(b));   \
        }                                                                                                 \
        const uint64_t t1 = memtime();                                                                    \
        uint64_t s = 0; for (int i = 0; i < 8; i++) s += a[i] + x[i];                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(s ^ (s >> 32));                           \
        if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;               \
    }
OPX(k_mad_u64_u32, "v_mad_u64_u32 %0, s[20:21], %1, %2, %0")
OPX(k_cvt_f64_u32, "v_cvt_f64_u32 %0, %1")

static double g_clock_ghz = 0.0;

template <typename F> double median_ticks(F kern, int blocks, uint64_t *d_ticks, uint32_t *d_out) {
    kern<<<blocks, 256>>>(d_ticks, d_out, 12345u);                 // warm (code fetch, clocks up)
    kern<<<blocks, 256>>>(d_ticks, d_out, 12345u);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    std::vector<uint64_t> h((size_t)blocks * 4);
    if (hipMemcpy(h.data(), d_ticks, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    std::sort(h.begin(), h.end());
    return (double)h[h.size() / 2];
}
template <typename F> void run(const char *name, F kern, uint64_t *d_ticks, uint32_t *d_out) {
    // 1 wave per SIMD: 256 blocks of 4 waves, one block per CU;  8 waves per SIMD: 2048 blocks
    const double t1 = median_ticks(kern, 256, d_ticks, d_out), t8 = median_ticks(kern, 2048, d_ticks, d_out);
    const double instr = (double)N * 8;                             // per wave
    const double c1 = t1 * 10.0 * g_clock_ghz / instr;              // cycles per instruction, this wave alone on its SIMD
    const double c8 = t8 * 10.0 * g_clock_ghz / (instr * 8);        // 8 waves share the SIMD: per instruction issued on it
    printf("%-18s  1 wave/SIMD: %6.2f cycles per instr     8 waves/SIMD: %6.2f cycles per instr per SIMD\n", name, c1, c8);
}

int main() {
    uint64_t *d_ticks; uint32_t *d_out;
    CHK(hipMalloc(&d_ticks, (size_t)2048 * 4 * 8));
    CHK(hipMalloc(&d_out, (size_t)2048 * 256 * 4));
    // ---- clock: N trips x (128 nop cycles + ~3) measured in 10 ns ticks, one wave per SIMD; cross-checked with HIP events
    //      around a long launch of the same kernel (ticks really are 10 ns)
    for (int rep = 0; rep < 3; rep++) {
        const double t = median_ticks(k_nop, 256, d_ticks, d_out);
        g_clock_ghz = (double)N * 128.0 / (t * 10.0);
        printf("clock from s_nop block: %.0f ticks of 10 ns for %d x 128 idle cycles -> %.3f GHz (lower bound, loop overhead < 3 %%)\n", t, N, g_clock_ghz);
    }
    {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a); k_nop<<<256, 256>>>(d_ticks, d_out, 1u); hipEventRecord(b); CHK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<uint64_t> h(1024); CHK(hipMemcpy(h.data(), d_ticks, 1024 * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        printf("s_memtime tick check: kernel %.1f us by HIP events, %.1f us by its waves' own ticks at 10 ns (median)\n", ms * 1e3, (double)h[512] * 0.01);
    }
#define R(k) run(#k, k, d_ticks, d_out);
    R(k_add_u32_e32) R(k_add_u32_e64) R(k_xor_b32_e32) R(k_xor_b32_e64) R(k_min_u32) R(k_cndmask)
    R(k_mul_f32_e32) R(k_mul_f32_e64) R(k_add_f32_e32) R(k_fmac_f32_e32) R(k_fma_f32)
    R(k_pk_fma_f32) R(k_pk_mul_f32) R(k_pk_add_f32)
    R(k_lshl_add) R(k_alignbit) R(k_bfe) R(k_perm) R(k_mul_lo) R(k_mad_u24) R(k_dpp_shr1) R(k_mad_u64_u32)
    R(k_fma_f64) R(k_mul_f64) R(k_min_f64) R(k_rcp_f64) R(k_trunc_f64) R(k_cvt_f64_u32) R(k_lshl_b64) R(k_cmp_u64)
    return 0;
}
