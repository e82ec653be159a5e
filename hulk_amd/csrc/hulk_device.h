// hulk_device.h — device-side helpers shared by the kernel files of libhulkhip (wave64 code for CDNA4):
// wave primitives, the minimap2 hash, the exact jump hash (go-jump), the flush decision of a batch.
#pragma once
#include "hulk_internal.h"

namespace hulk {
namespace {

constexpr uint64_t TAB_EMPTY = 0x00000000000000FFull;   // never a minimizer value (see k_minimizer_bin)
constexpr uint64_t X_NONE = ~0ull;
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void wave_sync() {
    // one wave owns its LDS region: program order is enough for the hardware, this stops the
    // compiler from moving LDS accesses across the point.
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u_zero(uint32_t v) {    // 0 where the source lane is invalid / row masked
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}
// wave-wide inclusive prefix sum with DPP only (6 VALU instructions); lane 63 ends with the total
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x) {
    x += dpp_u_zero<0x111, 0xf>(x);      // row_shr:1
    x += dpp_u_zero<0x112, 0xf>(x);      // row_shr:2
    x += dpp_u_zero<0x114, 0xf>(x);      // row_shr:4
    x += dpp_u_zero<0x118, 0xf>(x);      // row_shr:8      -> scan inside each row of 16
    x += dpp_u_zero<0x142, 0xa>(x);      // row_bcast15 -> rows 1,3
    x += dpp_u_zero<0x143, 0xc>(x);      // row_bcast31 -> rows 2,3
    return x;
}



// minimap2 hash64 — src/minimizer/minimizer.go:33-42
__device__ __forceinline__ uint64_t hash64(uint64_t key, uint64_t mask) {
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

// Jump consistent hash (Lamping & Veach) == go-jump Hash(key, n); fp64 divide and multiply are
// IEEE-exact on gfx950, so the result is bit-identical to the Go code.
// RN(1/r) for an integer 1 <= r <= 2^31 without the full IEEE division sequence: hardware
// reciprocal estimate + two FMA Newton steps.  With an exact residual e = 1 - r*y (FMA) the second
// step rounds correctly (Markstein); tests/test_gpu_parity.py::test_reciprocal_exhaustive checks
// every r in [1, 2^31] against IEEE division on the device.  2^31/r = 2^31 * RN(1/r) exactly.
__device__ __forceinline__ double rcp_exact_u31(uint32_t r) {
    const double d = (double)r;
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    return y;
}

// The same quotient the reference forms: float64(1<<31) / float64(r) = RN(2^31 / r).  r * 2^-31 is built by
// lowering the exponent field of float64(r) (an integer add on the upper dword), the Newton iteration is the
// scaled image of rcp_exact_u31's; k_selftest_rcp checks every r in [1, 2^31] against the IEEE quotient too.
__device__ __forceinline__ double quot31_exact(uint32_t r) {
    uint64_t bits = (uint64_t)__double_as_longlong((double)r) - (31ull << 52);
    const double d = __longlong_as_double((long long)bits);          // r * 2^-31, exact
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    return y;
}

__device__ __forceinline__ int32_t jump_hash(uint64_t key, int32_t n) {
    // b+1 <= n < 2^31 and (key>>33)+1 <= 2^31 convert exactly from uint32; the product is only
    // needed (a) to decide j >= n and (b), when j < n, as a value below 2^31 — so the int64
    // conversion of the Go code is replaced by a double compare + an exact int32 truncation.
    // float64(b+1) * (2^31 / r) == 2^31 * fl(float64(b+1) * RN(1/r))  (power-of-two scaling is exact)
    const double dn = (double)n * 0x1p-31;
    int32_t res = 0;
    uint32_t j = 0;
    for (;;) {
        res = (int32_t)j;                 // b = j
        key = key * 2862933555777941757ull + 1;
        const double p = (double)(j + 1u) * rcp_exact_u31((uint32_t)(key >> 33) + 1u);
        if (p >= dn) break;               // j >= n
        j = (uint32_t)(int32_t)(p * 0x1p31);   // trunc, exact (< n < 2^31)
    }
    return res;
}

// spectrum (ring slot) a read is binned into: interval rule of pipeline/sketch.go:211
__device__ __forceinline__ uint32_t hist_slot(const MinimizerParams &P, uint64_t rd) {
    if (P.interval == 0) return P.ring_base;
    return (uint32_t)(((P.fill + rd) / P.interval + P.ring_base) % P.ring_n);
}

__device__ __forceinline__ void set_error(DevState *st, int code) { atomicCAS(&st->err, 0, code); }

// ------------------------------------------------------------------------------------------
// Flush kernels.  A flush covers `count` consecutive spectra of the ring (FlushBatch), i.e. up to
// SCAN_BATCH sketching intervals at once: K2/K3 run per spectrum, K4a streams the K table ONCE
// for all of them, K4b applies the updates in interval order.  The result is identical to
// flushing the intervals one by one (per slot a running arg-min in stream order).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ring_slot(const FlushBatch &fb, int t) { return (fb.ring_base + (uint32_t)t) % fb.ring_n; }

// Flush decision per spectrum: boss.go:118 (skip empty spectrum) and kmerspectrum.go:88-96
// (fatal below 1 % used bins).
__device__ __forceinline__ bool flush_go(const DevState *st, const FlushBatch &fb, int t) {
    const unsigned used = st->used[fb.parity][ring_slot(fb, t)];
    if (used == 0) return false;
    const double prop = (double)used / (double)fb.num_bins;
    return !(prop < 0.01);
}

// all flush decisions of the batch with ONE memory round trip (call with the whole wave active)
__device__ __forceinline__ uint32_t batch_gomask(const DevState *st, const FlushBatch &fb) {
    const int lane = lane_id();
    bool go = false;
    if (lane < (int)fb.count) go = flush_go(st, fb, lane);
    return (uint32_t)__ballot(go);
}

}  // namespace
}  // namespace hulk
