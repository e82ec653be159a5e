// Second, independent measurement of the VALU issue cost on gfx950 (VERDICT r02 "corroborate or retire the 4-cycle
// claim"): op_cost.hip times whole launches with HIP events and divides by a NOMINAL 2.4 GHz.  Here every wave reads
// s_memtime before and after its unrolled block and records which SIMD it ran on (HW_REG_HW_ID, HW_REG_XCC_ID).  On this
// chip s_memtime advances at the shader clock (its rate is printed per kernel against HIP-event time: it follows the
// clock down under load), so the figures are CYCLES, with no clock assumed:
//   per SIMD:  instructions issued by all its waves / (last wave's end - first wave's start)      [8 waves per SIMD asked]
//   per wave:  its own ticks / its own instructions                                               [1 wave per SIMD]
// Each op in its VOP2 (32-bit encoding) and VOP3 (_e64) form where both exist, plus the packed fp32 ops.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <map>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int N = 4096;       // loop trips; 8 chains each

__device__ __forceinline__ uint64_t memtime() {
    uint64_t t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

__device__ __forceinline__ void record(uint64_t *ticks, uint64_t t0, uint64_t t1) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
    if ((threadIdx.x & 63) == 0) {
        uint64_t *o = ticks + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 3;
        o[0] = t0; o[1] = t1; o[2] = ((uint64_t)(xcc & 0xf) << 32) | hw;
    }
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
        record(ticks, t0, t1);               \
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
        record(ticks, t0, t1);               \
    }

// 8 x s_nop 15 per trip (what a wave's s_nop costs in ticks is printed; N * 8 of them)
__global__ __launch_bounds__(256) void k_nop(uint64_t *ticks, uint32_t *out, uint32_t seed) {
    const uint64_t t0 = memtime();
#pragma unroll 1
    for (int it = 0; it < N; it++) {
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");
    }
    const uint64_t t1 = memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = seed;
    record(ticks, t0, t1);
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
OP32(k_cmp_cndmask, "v_cmp_lt_u32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc")       // two instructions
OP32(k_and_b32, "v_and_b32_e32 %0, %0, %1")
OP32(k_or_b32, "v_or_b32_e32 %0, %0, %1")
OP32(k_lshlrev_b32, "v_lshlrev_b32_e32 %0, 3, %0")
OP32(k_lshrrev_b32, "v_lshrrev_b32_e32 %0, 3, %0")
OP32(k_sub_u32, "v_sub_u32_e32 %0, %0, %1")
OP32(k_mov_b32, "v_mov_b32_e32 %0, %1")
OP32(k_max_u32, "v_max_u32_e32 %0, %0, %1")
OP32(k_min_f32, "v_min_f32_e32 %0, %0, %1")
OP32(k_mul_u32_u24, "v_mul_u32_u24_e32 %0, %0, %1")
OP32(k_add_co_u32, "v_add_co_u32_e32 %0, vcc, %0, %1")
OP32(k_addc_co_u32, "v_addc_co_u32_e32 %0, vcc, %0, %1, vcc")
OP32(k_add3_u32, "v_add3_u32 %0, %0, %1, %1")
OP32(k_and_or_b32, "v_and_or_b32 %0, %0, %1, %1")
OP32(k_lshl_or_b32, "v_lshl_or_b32 %0, %0, 3, %1")
OP32(k_xad_u32, "v_xad_u32 %0, %0, %1, %1")
OP32(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
OP32(k_cvt_f32_u32, "v_cvt_f32_u32_e32 %0, %0")
OP32(k_rcp_f32, "v_rcp_f32_e32 %0, %0")
OP32(k_readlane_add, "v_readfirstlane_b32 s20, %0\n\tv_add_u32_e32 %0, s20, %1")                   // two instructions
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
        record(ticks, t0, t1);               \
    }
OPX(k_mad_u64_u32, "v_mad_u64_u32 %0, s[20:21], %1, %2, %0")
OPX(k_cvt_f64_u32, "v_cvt_f64_u32 %0, %1")

struct Result { double cyc_per_instr_simd, waves_per_simd, residency, tick_ghz, cyc_per_instr_wave; };

template <typename F> Result measure(F kern, int blocks, uint64_t *d_ticks, uint32_t *d_out) {
    Result r{};
    kern<<<blocks, 256>>>(d_ticks, d_out, 12345u);                 // warm (code fetch, clocks up)
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); kern<<<blocks, 256>>>(d_ticks, d_out, 12345u); hipEventRecord(b);
    if (hipDeviceSynchronize() != hipSuccess) return r;
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const size_t nw = (size_t)blocks * 4;
    std::vector<uint64_t> h(nw * 3);
    if (hipMemcpy(h.data(), d_ticks, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return r;
    // SIMD key: XCC_ID | SE_ID[15:13] SH_ID[12] CU_ID[11:8] SIMD_ID[5:4] of HW_ID
    std::map<uint64_t, std::vector<std::pair<uint64_t, uint64_t>>> simd;
    uint64_t lo = ~0ull, hi = 0; std::vector<double> per_wave;
    for (size_t w = 0; w < nw; w++) {
        const uint64_t t0 = h[w * 3], t1 = h[w * 3 + 1], id = h[w * 3 + 2];
        simd[(id >> 32 << 32) | (id & 0xff30u)].push_back({t0, t1});
        lo = std::min(lo, t0); hi = std::max(hi, t1);
        per_wave.push_back((double)(t1 - t0) / ((double)N * 8));
    }
    std::vector<double> c, res, cnt;
    for (auto &kv : simd) {
        uint64_t s0 = ~0ull, s1 = 0; double busy = 0;
        for (auto &p : kv.second) { s0 = std::min(s0, p.first); s1 = std::max(s1, p.second); busy += (double)(p.second - p.first); }
        c.push_back((double)(s1 - s0) / ((double)kv.second.size() * N * 8));
        res.push_back(busy / (double)(s1 - s0)); cnt.push_back((double)kv.second.size());
    }
    auto med = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    r.cyc_per_instr_simd = med(c); r.waves_per_simd = med(cnt); r.residency = med(res);
    r.cyc_per_instr_wave = med(per_wave);
    {   // s_memtime is per XCD (unsynchronised offsets): its rate = a SIMD's own window against the launch's HIP-event time
        std::vector<double> wnd;
        for (auto &kv : simd) { uint64_t s0 = ~0ull, s1 = 0; for (auto &p : kv.second) { s0 = std::min(s0, p.first); s1 = std::max(s1, p.second); } wnd.push_back((double)(s1 - s0)); }
        r.tick_ghz = med(wnd) / (ms * 1e6);
    }
    (void)lo; (void)hi;
    return r;
}
template <typename F> void run(const char *name, F kern, uint64_t *d_ticks, uint32_t *d_out) {
    const Result one = measure(kern, 256, d_ticks, d_out), eight = measure(kern, 2048, d_ticks, d_out);
    printf("%-16s 1 wave/SIMD: %5.2f cyc/instr (wave)   | %2.0f waves/SIMD (%.1f resident on average): %5.2f cyc/instr per SIMD, "
           "a wave sees %5.2f; clock >= %.2f GHz (1 wave) / %.2f (8)\n", name, one.cyc_per_instr_wave, eight.waves_per_simd, eight.residency,
           eight.cyc_per_instr_simd, eight.cyc_per_instr_wave, one.tick_ghz, eight.tick_ghz);
}

int main() {
    uint64_t *d_ticks; uint32_t *d_out;
    CHK(hipMalloc(&d_ticks, (size_t)2048 * 4 * 3 * 8));
    CHK(hipMalloc(&d_out, (size_t)2048 * 256 * 4));
    {   // what an s_nop costs, and the rate of s_memtime against HIP-event time on an idle-ish chip
        const Result r = measure(k_nop, 256, d_ticks, d_out);
        printf("k_nop: s_nop 15 = %.1f ticks each (1 wave per SIMD: 16 issue slots of 4 cycles + the loop); s_memtime advances %.3f ticks per ns of\n"
               "       HIP-event time of the launch (a lower bound: the event time includes the launch overhead) = the shader clock\n",
               r.cyc_per_instr_wave, r.tick_ghz);
    }
#define R(k) run(#k, k, d_ticks, d_out);
    R(k_add_u32_e32) R(k_add_u32_e64) R(k_sub_u32) R(k_xor_b32_e32) R(k_xor_b32_e64) R(k_and_b32) R(k_or_b32) R(k_mov_b32)
    R(k_lshlrev_b32) R(k_lshrrev_b32) R(k_min_u32) R(k_max_u32) R(k_min_f32) R(k_mul_u32_u24) R(k_add_co_u32) R(k_addc_co_u32)
    R(k_cmp_cndmask) R(k_readlane_add) R(k_add3_u32) R(k_and_or_b32) R(k_lshl_or_b32) R(k_xad_u32) R(k_bcnt) R(k_cvt_f32_u32) R(k_rcp_f32)
    R(k_mul_f32_e32) R(k_mul_f32_e64) R(k_add_f32_e32) R(k_fmac_f32_e32) R(k_fma_f32)
    R(k_pk_fma_f32) R(k_pk_mul_f32) R(k_pk_add_f32)
    R(k_lshl_add) R(k_alignbit) R(k_bfe) R(k_perm) R(k_mul_lo) R(k_mad_u24) R(k_dpp_shr1) R(k_mad_u64_u32)
    R(k_fma_f64) R(k_mul_f64) R(k_min_f64) R(k_rcp_f64) R(k_trunc_f64) R(k_cvt_f64_u32) R(k_lshl_b64) R(k_cmp_u64)
    return 0;
}
