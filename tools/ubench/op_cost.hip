// Micro-benchmark: issue cost of single gfx950 VALU instructions, one inline-asm instruction per kind, 8 independent
// register chains per lane and 8 waves per SIMD (throughput, not latency).  Prints cycles per wave64 instruction per SIMD.
// These are the figures behind the per-step budgets of k_jump_bin / k_minimizer_fast in docs/EXPERIMENTS.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int N = 2048;

// 32-bit destination, two 32-bit sources
#define OP32(NAME, ASM)                                                                                   \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed) {                           \
        uint32_t a[8];                                                                                    \
        for (int i = 0; i < 8; i++) a[i] = seed * (i + 3) + threadIdx.x;                                   \
        uint32_t b = seed | 1u;                                                                           \
        _Pragma("unroll 1") for (int it = 0; it < N; it++) {                                              \
            _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b));       \
        }                                                                                                 \
        uint32_t s = 0; for (int i = 0; i < 8; i++) s += a[i];                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                   \
    }
// 64-bit destination chain
#define OP64(NAME, ASM)                                                                                   \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed) {                           \
        uint64_t a[8];                                                                                    \
        for (int i = 0; i < 8; i++) a[i] = 0x3ff0000000000000ull + ((uint64_t)(seed * (i + 3) + threadIdx.x) << 20);   \
        uint64_t b = 0x3ff0000000000123ull + seed;                                                        \
        _Pragma("unroll 1") for (int it = 0; it < N; it++) {                                              \
            _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b));       \
        }                                                                                                 \
        uint64_t s = 0; for (int i = 0; i < 8; i++) s += a[i];                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(s ^ (s >> 32));                           \
    }

OP32(k_add_u32, "v_add_u32 %0, %0, %1")
OP32(k_xor_b32, "v_xor_b32 %0, %0, %1")
OP32(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
OP32(k_add3, "v_add3_u32 %0, %0, %1, %1")
OP32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7")
OP32(k_bfe, "v_bfe_u32 %0, %0, 3, 20")
OP32(k_perm, "v_perm_b32 %0, %0, %1, %1")
OP32(k_and_or, "v_and_or_b32 %0, %0, %1, %1")
OP32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
OP32(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
OP32(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %1")
OP32(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
OP32(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
OP32(k_cvt_u32_f32, "v_cvt_u32_f32 %0, %0")
OP32(k_rcp_f32, "v_rcp_f32 %0, %0")
OP32(k_fma_f32, "v_fma_f32 %0, %0, %1, %1")
OP32(k_mul_f32, "v_mul_f32 %0, %0, %1")
OP32(k_floor_f32, "v_floor_f32 %0, %0")
OP32(k_cmp_cnd, "v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc")
OP32(k_dpp_shr1, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
OP32(k_min_u32, "v_min_u32 %0, %0, %1")
OP32(k_min3_u32, "v_min3_u32 %0, %0, %1, %1")
OP32(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
OP32(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %0, %1")
OP32(k_readlane_like, "v_readfirstlane_b32 s20, %0\n\tv_add_u32 %0, s20, %1")
// 64-bit accumulator, 32-bit factors
#define OPX(NAME, ASM)                                                                                    \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed) {                           \
        uint64_t a[8]; uint32_t x[8];                                                                     \
        for (int i = 0; i < 8; i++) { a[i] = 0x3ff0000000000000ull + ((uint64_t)(seed * (i + 3) + threadIdx.x) << 20); x[i] = seed * i + threadIdx.x; } \
        uint32_t b = seed | 1u;                                                                           \
        _Pragma("unroll 1") for (int it = 0; it < N; it++) {                                              \
            _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(a[i]), "+v"(x[i]) : "v"(b));   \
        }                                                                                                 \
        uint64_t s = 0; for (int i = 0; i < 8; i++) s += a[i] + x[i];                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(s ^ (s >> 32));                           \
    }
OPX(k_mad_u64_u32, "v_mad_u64_u32 %0, s[20:21], %1, %2, %0")
OP64(k_fma_f64, "v_fma_f64 %0, %0, %1, %1")
OP64(k_mul_f64, "v_mul_f64 %0, %0, %1")
OP64(k_add_f64, "v_add_f64 %0, %0, %1")
OP64(k_min_f64, "v_min_f64 %0, %0, %1")
OP64(k_rcp_f64, "v_rcp_f64 %0, %0")
OP64(k_trunc_f64, "v_trunc_f64 %0, %0")
OP64(k_fract_f64, "v_fract_f64 %0, %0")
OPX(k_cvt_f64_u32, "v_cvt_f64_u32 %0, %1")
OPX(k_cvt_u32_f64, "v_cvt_u32_f64 %1, %0")
OP64(k_lshl_b64, "v_lshlrev_b64 %0, 3, %0")
OP64(k_lshr_b64, "v_lshrrev_b64 %0, 3, %0")
OP64(k_cmp_u64, "v_cmp_lt_u64 vcc, %0, %1")
OP64(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %1")
OP64(k_pk_mov, "v_pk_mov_b32 %0, %0, %1")
OP64(k_mov_b64, "v_mov_b64 %0, %1")

template <typename F> int run(const char *name, F kern, int instr_per_op, uint32_t *d) {
    const int blocks = 256 * 8;          // 8 waves per SIMD
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    kern<<<blocks, 256>>>(d, 12345u);
    hipEventRecord(a); kern<<<blocks, 256>>>(d, 12345u); hipEventRecord(b); CHK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ops_per_simd = (double)blocks * 4 / 1024 * N * 8;
    const double ns = ms * 1e6 / ops_per_simd;
    printf("%-28s %8.1f us  %6.2f cycles @2.4GHz per wave-op per SIMD  (%d instr per op)\n", name, ms * 1000, ns * 2.4, instr_per_op);
    return 0;
}
int main() {
    uint32_t *d; CHK(hipMalloc(&d, (size_t)256 * 8 * 256 * 4));
#define R(k, n) run(#k, k, n, d);
    R(k_add_u32, 1) R(k_xor_b32, 1) R(k_lshl_add, 1) R(k_add3, 1) R(k_alignbit, 1) R(k_bfe, 1) R(k_perm, 1) R(k_and_or, 1)
    R(k_min_u32, 1) R(k_min3_u32, 1) R(k_bcnt, 1) R(k_mbcnt, 1) R(k_cmp_cnd, 2) R(k_dpp_shr1, 1) R(k_readlane_like, 2)
    R(k_mul_lo, 1) R(k_mul_hi, 1) R(k_mad_u24, 1) R(k_mul_u24, 1) R(k_mad_u64_u32, 1)
    R(k_cvt_f32_u32, 1) R(k_cvt_u32_f32, 1) R(k_rcp_f32, 1) R(k_fma_f32, 1) R(k_mul_f32, 1) R(k_floor_f32, 1)
    R(k_fma_f64, 1) R(k_mul_f64, 1) R(k_add_f64, 1) R(k_min_f64, 1) R(k_rcp_f64, 1) R(k_trunc_f64, 1) R(k_fract_f64, 1)
    R(k_cvt_f64_u32, 1) R(k_cvt_u32_f64, 1) R(k_lshl_b64, 1) R(k_lshr_b64, 1) R(k_cmp_u64, 1) R(k_cmp_f64, 1) R(k_pk_mov, 1) R(k_mov_b64, 1)
    return 0;
}
