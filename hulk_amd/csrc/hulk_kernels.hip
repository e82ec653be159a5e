// hulk_kernels.hip — gfx950 kernels of the HULK `sketch` hot path (DESIGN.md §3 describes each).
//
//   K1a k_minimizer_fast   short reads -> distinct minimizers per read (list in HBM)
//   K1b k_jump_bin         jump hash of the list                  (kmerspectrum.go:67-81, go-jump)
//   K1c k_range_hist/k_merge_hist   k-mer spectrum in LDS, no global atomics
//       k_minimizer_bin    reads of up to 1024 k-mer positions, any bytes (fused jump hash + atomics)
//       k_long_hash/k_long_emit   long reads and contigs, grouped launches
//                          (reference: src/minimizer/minimizer.go:96-204)
//   K2  k_count_used       KmerSpectrum.Cardinality(), the 1 % rule       (kmerspectrum.go:53-55,84-96)
//       k_flush_decide     whole-batch bound: can any element still change the sketch?
//   K3  k_cms_segsum/k_cms_base/k_cms_freq   count-min Add() for a batch of spectra, bin order
//       k_cmsd_*           the same with uniform scaling (src/countmin/countmin.go:103-147)
//   K4  k_cws_scan         fp32 pass over K = c*exp(b-r): per (interval, slot, tile) minimum, bound-pruned
//       k_cws_resolve(+_drift)/k_cws_apply   exact fp64 re-evaluation with the literal formula of
//                          src/histosketch/histosketch.go:30-33 and the slot update (histosketch.go:135-153)
//       k_cws_eval/k_cws_scatter/k_cws_beta/k_build_k32   the CWS tables (histosketch.go:95-126)
//       k_smash            pairwise distances of `hulk smash`
//
// All kernels are wave64 code for CDNA4; none of them has a CPU or library fallback.
#include "hulk_internal.h"

#include <math.h>
#include <stdlib.h>
#include <algorithm>

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
// K1: minimizers + binning.  One wave owns a read at a time.
//
// LDS per wave:  Xs[xcap] u64   hashed k-mer (or X_NONE) per k-mer position
//                vm[xcap/64] u64 validity masks (position not skipped)
//                tab[tab_size] u64  open-addressing set = the per-read golang-set
//                q[128] u64     distinct minimizers waiting for a full-wave jump-hash pass
//                pk[...] u8     2-bit packed bases, base p at bits 2(p%4) of byte p/4
// LDS per block: lut[256]       seq_nt4_table (minimizer.go:13-30)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint8_t nt4_of(unsigned c) {
    unsigned u = c | 0x20u;
    if (c < 4) return (uint8_t)c;
    if (u == 'a') return 0;
    if (u == 'c') return 1;
    if (u == 'g') return 2;
    if (u == 't' || u == 'u') return 3;
    return 4;
}

__global__ __launch_bounds__(256) void k_minimizer_bin(const uint8_t *__restrict__ bases,
                                                       const uint64_t *__restrict__ offsets,
                                                       uint64_t n_reads, MinimizerParams P,
                                                       uint32_t *__restrict__ hist, DevState *st,
                                                       unsigned long long *__restrict__ min_slots,
                                                       const uint32_t *__restrict__ read_list,
                                                       const uint32_t *__restrict__ read_list_count) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint8_t *lut = smem;
    if (read_list) n_reads = *read_list_count;      // second pass over the reads the fast kernel deferred
    for (int t = threadIdx.x; t < 256; t += blockDim.x) lut[t] = nt4_of((unsigned)t);
    __syncthreads();

    const int lane = lane_id();
    const int wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t xcap = P.xcap, tabn = P.tab_size, tabmask = P.tab_size - 1;
    const size_t per_wave = P.lds_per_wave;
    unsigned char *wbase = smem + 256 + (size_t)wid * per_wave;
    uint64_t *Xs = (uint64_t *)wbase;
    uint64_t *vm = Xs + xcap;
    uint64_t *tab = vm + (xcap + 63) / 64;
    uint64_t *q = tab + tabn;
    uint32_t *qs = (uint32_t *)(q + 128);                      // spectrum slot of each queued value
    uint8_t *pk8 = (uint8_t *)(q + 128 + 64);
    const uint32_t *pk32 = (const uint32_t *)pk8;

    const int32_t k = (int32_t)P.k, w = (int32_t)P.w;
    const int32_t wwin = w > 0 ? w : 1;   // w == 0: the deque is emptied every step, same window as w == 1
    const uint64_t mask = (1ull << (2 * k)) - 1;
    const uint64_t shift = (uint64_t)(2 * (k - 1));

    for (uint32_t s = lane; s < tabn; s += 64) tab[s] = TAB_EMPTY;
    wave_sync();

    uint32_t qn = 0;                 // wave-uniform
    unsigned long long nmin = 0;     // wave-uniform
    const uint32_t dbg = P.debug;    // ablation switches for tools/k1_ablate.py (0 in production)
    uint32_t sink = 0;

    if (!read_list && blockIdx.x == 0 && threadIdx.x == 0 && n_reads)
        atomicAdd(&st->total_len, (unsigned long long)(offsets[n_reads] - offsets[0]));

    const uint64_t gw = (uint64_t)blockIdx.x * nw + wid, stride = (uint64_t)gridDim.x * nw;
    for (uint64_t ri = gw; ri < n_reads; ri += stride) {
        const uint64_t rd = read_list ? (uint64_t)read_list[ri] : ri;
        const uint32_t hslot = hist_slot(P, rd);             // which k-mer spectrum of the ring
        const uint64_t o0 = offsets[rd], o1 = offsets[rd + 1];
        const int64_t L = (int64_t)(o1 - o0);
        // NewMinimizerSketch checks (minimizer.go:70-76); errors are deferred to hulk_finish
        if (L < 1) { if (lane == 0) set_error(st, -3); continue; }
        if (L < (int64_t)(w + k - 1)) { if (lane == 0) set_error(st, -4); continue; }
        const int64_t npos64 = L - k + 1;
        if (npos64 > (int64_t)xcap) { if (lane == 0 && !P.skip_long) set_error(st, -33); continue; }
        const int32_t npos = (int32_t)npos64;

        // ---- stage: ASCII -> 2-bit packs in LDS (4 bases per lane per pass), detect code 4
        bool sawN = false;
        for (int64_t b0 = 0; b0 < L; b0 += 256) {
            const int64_t p = b0 + 4 * lane;
            const int64_t left = L - p;
            if (left > 0) {
                const uintptr_t addr = (uintptr_t)(bases + o0 + (uint64_t)p);
                const uintptr_t al = addr & ~(uintptr_t)3;
                const unsigned sh = (unsigned)(addr & 3) * 8;
                const uint32_t lo = *(const uint32_t *)al;       // aligned dword holding base p
                uint32_t hi = 0;
                if (sh && al + 8 <= (uintptr_t)bases + P.bases_bytes) hi = *(const uint32_t *)(al + 4);
                const uint32_t by = sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
                const int nv = left < 4 ? (int)left : 4;
                unsigned pack = 0;
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    unsigned c = lut[(by >> (8 * t)) & 0xff];
                    if (t < nv) { sawN |= (c > 3); pack |= (c & 3u) << (2 * t); }
                }
                pk8[p >> 2] = (uint8_t)pack;
            }
        }
        const bool hasN = __ballot(sawN) != 0ull;
        wave_sync();

        // ---- hashed canonical k-mer per position (minimizer.go:126-159)
        for (int32_t j0 = 0; j0 < npos; j0 += 64) {
            const int32_t j = j0 + lane;          // first base of the k-mer
            const int32_t i = j + k - 1;          // its last base = the reference's loop index
            uint64_t X = X_NONE;
            bool valid = false;
            if (j < npos) {
                uint64_t f, r;
                if (!hasN) {
                    const uint32_t bo = 2u * (uint32_t)j, d = bo >> 5, o = bo & 31u;
                    const uint64_t lo = (uint64_t)pk32[d] | ((uint64_t)pk32[d + 1] << 32);
                    uint64_t W = o ? (lo >> o) | ((uint64_t)pk32[d + 2] << (64 - o)) : lo;
                    W &= mask;                    // base j at bits 0..1, base i at bits 2(k-1)..
                    uint64_t rev = __brevll(W) >> (64 - 2 * k);
                    f = ((rev >> 1) & 0x5555555555555555ull) | ((rev & 0x5555555555555555ull) << 1);
                    r = (~W) & mask;
                } else {
                    // literal recurrence; bases before i-k cannot reach bit positions that
                    // survive (f is masked every step, r loses 2 bits per step)
                    f = 0; r = 0;
                    int32_t p0 = i - k; if (p0 < 0) p0 = 0;
                    for (int32_t p = p0; p <= i; p++) {
                        const uint64_t c = lut[bases[o0 + (uint64_t)p]];
                        f = (f << 2 | c) & mask;
                        r = (r >> 2) | ((3ull ^ c) << shift);
                    }
                }
                if (f != r) {
                    const uint64_t canon = f > r ? r : f;
                    int32_t span = i - w + 2;     // windowIndex + 1
                    if (span >= k) span = k;
                    X = hash64(canon, mask) << 8 | (uint64_t)(int64_t)span;
                    valid = true;
                }
                Xs[j] = X;
            }
            const uint64_t vmask = __ballot(valid);
            if (lane == 0) vm[j0 >> 6] = vmask;
        }
        wave_sync();

        // ---- windowed minimum, per-read set insert, queue new values (minimizer.go:162-199)
        uint64_t carry_m = 0; bool carry_emit = false;     // wave-uniform: last lane of previous pass
        for (int32_t j0 = 0; j0 < npos; j0 += 64) {
            const int32_t j = j0 + lane;
            const int32_t i = j + k - 1;
            const uint64_t vmask = vm[j0 >> 6];
            const bool emit = (j < npos) && ((vmask >> lane) & 1ull) && (i >= w - 1);
            uint64_t m = X_NONE;
            if (emit) {
                int32_t lo = j - (wwin - 1); if (lo < 0) lo = 0;
                for (int32_t p = lo; p <= j; p++) { const uint64_t x = Xs[p]; m = x < m ? x : m; }
            }
            uint64_t pm = __shfl_up(m, 1);
            int pe = __shfl_up((int)emit, 1);
            if (lane == 0) { pm = carry_m; pe = (int)carry_emit; }
            carry_m = __shfl(m, 63); carry_emit = __shfl((int)emit, 63) != 0;
            const bool start = emit && !(pe && pm == m);
            bool isnew = false;
            if (dbg & 8u) { sink += (uint32_t)m; } else
            if (dbg & 4u) { isnew = start; } else
            if (start) {
                uint32_t slot = ((uint32_t)(m >> 8) ^ (uint32_t)(m >> 37)) & tabmask;
                for (;;) {
                    const unsigned long long old =
                        atomicCAS((unsigned long long *)&tab[slot], (unsigned long long)TAB_EMPTY,
                                  (unsigned long long)m);
                    if (old == TAB_EMPTY) { isnew = true; break; }
                    if (old == m) break;
                    slot = (slot + 1) & tabmask;
                }
            }
            const uint64_t nb = __ballot(isnew);
            if (nb) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(nb >> 32),
                                          __builtin_amdgcn_mbcnt_lo((uint32_t)nb, 0u));
                if (isnew) { q[qn + rank] = m; qs[qn + rank] = hslot; }
                qn += (uint32_t)__popcll(nb);
                wave_sync();
            }
            if (qn >= 64) {
                // full-wave jump-hash pass (kmerspectrum.go:70,78)
                const uint64_t x = q[lane];
                const uint32_t xs = qs[lane];
                const uint64_t keep = (lane + 64u < qn) ? q[lane + 64] : 0;
                const uint32_t keeps = (lane + 64u < qn) ? qs[lane + 64] : 0;
                const int32_t bin = (dbg & 2u) ? (int32_t)((uint32_t)(x >> 20) & 0xffffu) : jump_hash(x, P.num_bins);
                if (dbg & 1u) sink += (uint32_t)bin; else
                atomicAdd(&hist[(size_t)xs * (size_t)P.num_bins + bin], 1u);
                wave_sync();
                q[lane] = keep; qs[lane] = keeps;
                wave_sync();
                qn -= 64; nmin += 64;
            }
        }
        // clear the per-read set
        for (uint32_t s = lane; s < tabn; s += 64) tab[s] = TAB_EMPTY;
        wave_sync();
    }
    if (qn) {
        if ((uint32_t)lane < qn)
            atomicAdd(&hist[(size_t)qs[lane] * (size_t)P.num_bins + jump_hash(q[lane], P.num_bins)], 1u);
        nmin += qn;
    }
    // same-address atomics serialise at ~12 ns each on this chip: every block owns one slot of
    // min_slots[] instead (launches are stream-ordered, so a plain read-modify-write is safe)
    __shared__ unsigned long long blk_nmin[4];
    if (lane == 0) blk_nmin[wid] = nmin;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int x = 0; x < nw; x++) t += blk_nmin[x];
        if (t) min_slots[blockIdx.x] += t;
    }
    if (dbg && sink == 0xdeadbeefu) hist[0] = sink;     // keep ablated work alive
}


// ------------------------------------------------------------------------------------------
// K1-fast: the short-read form of K1.  A 16-lane group (one DPP row) owns a read; lane g of the
// group owns the block of w consecutive k-mer positions [g*w, (g+1)*w) and walks it with the
// rolling 2-bit k-mers of the reference (one extraction from the packed read, then shift-in per
// base), so the minimap2 hash is the only per-position cost; the block is fully unrolled (WM >= w
// register slots) so the w independent hash chains interleave.  With blocks of exactly w
// positions the windowed minimum is the van Herk/Gil-Werman form: min(suffix-min of the previous
// block — fetched from the neighbouring lane with DPP row_shr:1 —, prefix-min of the own block).
// Per-read set semantics: a 128-entry open-addressing set per group; all run-start values of a
// lane are inserted with back-to-back LDS compare-and-swaps (one round trip), collisions probe on.
// Eligible reads: no code-4 base, 1 <= w <= WM <= 16, k-mer positions <= 16*w, length <= 256,
// <= 64 run starts.  Anything else is appended to slow_list and handled by k_minimizer_bin.
//
// LDS per group: tab[128] u64 | pk[20] u32      per wave: q[192] u64      per block: lut[256]
// ------------------------------------------------------------------------------------------
constexpr int FAST_TAB = 128;
constexpr int FAST_CAND = 64;          // max run starts per read on the fast path
constexpr int FAST_RAW = 3072 + 64;    // raw ASCII of the wave's 16 reads, staged once (bytes per wave)
constexpr int FAST_RAW_PAIR = 5120 + 64;   // ... when two groups share a read (reads of up to ~300 bases)

__device__ __forceinline__ uint32_t dpp_row_shr1(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
}
__device__ __forceinline__ uint64_t dpp_row_shr1_u64(uint64_t v) {
    return (uint64_t)dpp_row_shr1((uint32_t)v) | ((uint64_t)dpp_row_shr1((uint32_t)(v >> 32)) << 32);
}

// hash64 for 2k <= 54 (FM kernels): the upper dword then has at most 22 bits, so the two multiply steps
// (x265, x21) take v_mad_u64_u32 for the low dword and ONE full-rate v_mad_u32_u24 for the upper one,
// instead of two v_mad_u64_u32 with a v_mov between them.  Same value as hash64 (mod 2^2k).
// 32 x 32 -> 64 multiply as ONE v_mad_u64_u32.  In C++ hipcc folds the upper-dword addend into the mad, feeds it
// through a v_mov, and then recomputes the product's low dword with a second v_mul_lo_u32 for the next xor.
__device__ __forceinline__ uint64_t mul_u32_u64(uint32_t a, uint32_t b) {
    uint64_t d, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(carry) : "v"(a), "s"(b));
    return d;
}
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c));
    return d;
}
template <bool FM> __device__ __forceinline__ uint64_t hash64_fm(uint64_t key, uint64_t mask) {
    if (!FM) return hash64(key, mask);
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    {
        const uint64_t p = mul_u32_u64((uint32_t)key, 265u);
        const uint32_t hi = mad_u24((uint32_t)(key >> 32), 265u, (uint32_t)(p >> 32));
        key = (((uint64_t)hi << 32) | (uint32_t)p) & mask;
    }
    key = key ^ key >> 14;
    {
        const uint64_t p = mul_u32_u64((uint32_t)key, 21u);
        const uint32_t hi = mad_u24((uint32_t)(key >> 32), 21u, (uint32_t)(p >> 32));
        key = (((uint64_t)hi << 32) | (uint32_t)p) & mask;
    }
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

// hash64 of a 2KC-bit k-mer (17 <= KC <= 27: the upper dword has 2KC-32 <= 22 bits) and the packing
// X = hash << 8 | span, in explicit dword form: every "key +- key << n" step of the hash is a multiplication by a
// constant mod 2^2KC (x(2^21-1) - 1, x265, x21, x(2^31+1)) = ONE v_mad_u64_u32 on the low dword + one
// v_mad_u32_u24 / v_add on the upper one + its mask; every xor-shift is one v_alignbit + v_xor on the low dword
// (the upper dword shifted by 24 or 28 is zero).  20 instructions, 4 of them multiplies; hipcc's rendering of
// hash64() + the packing had 30 with 6 multiplies.
__device__ __forceinline__ uint64_t mad_u32_u64_m1(uint32_t a, uint32_t b) {
    uint64_t d, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, -1" : "=v"(d), "=s"(carry) : "v"(a), "s"(b));
    return d;
}
template <int KC> __device__ __forceinline__ uint64_t hash64_pack_kc(uint64_t key, uint32_t span) {
    static_assert(KC >= 17 && KC <= 27, "upper dword of the k-mer must have 2..22 bits");
    constexpr uint32_t HM = (1u << (2 * KC - 32)) - 1u;
    uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
    uint64_t p;
    p = mad_u32_u64_m1(lo, 0x1FFFFFu);                             // ~key + (key << 21) = key * (2^21 - 1) - 1
    hi = mad_u24(hi, 0x1FFFFFu, (uint32_t)(p >> 32)) & HM; lo = (uint32_t)p;
    lo ^= __builtin_amdgcn_alignbit(hi, lo, 24);                   // key ^= key >> 24
    p = mul_u32_u64(lo, 265u);                                     // key + (key << 3) + (key << 8)
    hi = mad_u24(hi, 265u, (uint32_t)(p >> 32)) & HM; lo = (uint32_t)p;
    lo ^= __builtin_amdgcn_alignbit(hi, lo, 14);                   // key ^= key >> 14
    if (2 * KC - 32 > 14) hi ^= hi >> 14;
    p = mul_u32_u64(lo, 21u);                                      // key + (key << 2) + (key << 4)
    hi = mad_u24(hi, 21u, (uint32_t)(p >> 32)) & HM; lo = (uint32_t)p;
    lo ^= __builtin_amdgcn_alignbit(hi, lo, 28);                   // key ^= key >> 28
    p = mul_u32_u64(lo, 0x80000001u);                              // key + (key << 31)
    hi = ((uint32_t)(p >> 32) + hi) & HM; lo = (uint32_t)p;
    const uint32_t xh = __builtin_amdgcn_alignbit(hi, lo, 24), xl = (lo << 8) | span;   // << 8 | span
    return ((uint64_t)xh << 32) | xl;
}

// 64-bit unsigned minimum.  For k <= 27 every minimizer value (hash64 << 8 | span < 2^62) and every
// 2k-bit k-mer is the bit pattern of a non-negative finite double, whose order is the integer order, so
// v_min_f64 (denormals preserved, kernel descriptor float_denorm_mode_16_64 = 3) does in ONE instruction
// what v_cmp_lt_u64 + 2 v_cndmask do in three; "no value" is then +inf (0x7FF0...) instead of ~0.
// Inline asm: the builtin would add a canonicalising v_max_f64 per operand.
template <bool FM> __device__ __forceinline__ uint64_t umin64(uint64_t a, uint64_t b) {
    if (FM) {
        double d;
        asm("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(__longlong_as_double((long long)a)), "v"(__longlong_as_double((long long)b)));
        return (uint64_t)__double_as_longlong(d);
    }
    return a < b ? a : b;
}

// PAIR: two neighbouring 16-lane groups share a read (2 x 16w - (w-1) k-mer positions: 300 bp at k = 21, w = 9).
// The second group starts w-1 positions before the first one ends, so that every window it reports is complete,
// and reports nothing for those w-1 positions; both groups feed the same per-read set.
template <int WM, bool FM, bool DBG, bool WEQ, int KC, bool PAIR>
__global__ __launch_bounds__(256, 4) void k_minimizer_fast(const uint8_t *__restrict__ bases,
                                                        const uint64_t *__restrict__ offsets,
                                                        uint64_t n_reads, MinimizerParams P,
                                                        MinimizerList ml, DevState *st,
                                                        unsigned long long *__restrict__ min_slots,
                                                        uint32_t *__restrict__ slow_list,
                                                        uint32_t *__restrict__ slow_count) {
    extern __shared__ __align__(16) unsigned char smem[];
    // (the first 2 KB of LDS held ASCII -> 2-bit tables once; the layout behind them is unchanged)

    constexpr uint64_t XN = FM ? 0x7FF0000000000000ull : X_NONE;   // "no value": above every minimizer value
    const int lane = lane_id(), wid = threadIdx.x >> 6;
    const int grp = threadIdx.x >> 4, gl = threadIdx.x & 15, gsh = lane & 48;
    // WEQ: the window size equals the block size WM (w = 9 is the reference's default): every `t < w` test and
    // every multiple of w becomes a compile-time constant
    const int32_t k = KC ? KC : (int32_t)P.k, w = WEQ ? WM : (int32_t)P.w;   // KC: k fixed at compile time (21 = the default)
    const uint64_t mask = (1ull << (2 * k)) - 1;
    const uint64_t shift = (uint64_t)(2 * (k - 1));
    constexpr int RAWB = PAIR ? FAST_RAW_PAIR : FAST_RAW;          // raw ASCII of the wave's 16 reads (bytes per wave)
    constexpr int RPI = PAIR ? 2 : 4;                              // reads per iteration of a wave
    constexpr bool DX = KC != 0 && WEQ && 2 * (KC + WM - 1) <= 64;     // direct k-mer extraction (phase A)
    constexpr bool HP = FM && WEQ && KC >= 17 && KC <= 27;                    // hash + packing in explicit dword form
    const int half = PAIR ? (grp & 1) : 0;                         // which half of the read this group takes
    const int sub = PAIR ? ((grp & 3) >> 1) : (grp & 3);           // read of the iteration
    const int32_t posoff = half ? 16 * w - (w - 1) : 0;            // first k-mer position of this group
    uint64_t *tab = (uint64_t *)(smem + 2048) + (size_t)(PAIR ? (grp & ~1) : grp) * FAST_TAB;   // the per-read set
    uint32_t *pk32 = (uint32_t *)(smem + 2048 + 16 * FAST_TAB * 8) + grp * 20;
    uint32_t *raw32 = (uint32_t *)(smem + 2048 + 16 * FAST_TAB * 8 + 16 * 20 * 4) + (size_t)wid * (RAWB / 4);
    uint64_t *cs = (uint64_t *)(smem + 2048 + 16 * FAST_TAB * 8 + 16 * 20 * 4 + 4 * RAWB) + (size_t)grp * FAST_CAND;
#pragma unroll
    for (int x = 0; x < FAST_TAB / 16; x++) tab[gl + 16 * x] = TAB_EMPTY;
    __syncthreads();

    // this wave owns FAST_READS_PER_WAVE consecutive reads and one region of the minimizer list
    const uint64_t region = (uint64_t)blockIdx.x * 4 + (uint64_t)wid;
    const uint64_t wave_first = region * FAST_READS_PER_WAVE;
    uint64_t *xl = ml.x + region * ml.rcap;
    uint8_t *sl8 = ml.slot + region * ml.rcap;
    uint32_t wcount = 0;              // wave-uniform: values written to the region so far
    // ablation switches (tools/k1_ablate.py) exist only in the DBG instantiation: in the production kernel they
    // cost a branch per k-mer position and SGPRs the compiler then spills to VGPR lanes
    const uint32_t dbg = DBG ? P.debug : 0u;
    uint32_t sink = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0 && n_reads)
        atomicAdd(&st->total_len, (unsigned long long)(offsets[n_reads] - offsets[0]));

    // ---- wave prologue: the 17 offsets of the wave's reads in ONE round trip, then (when the reads
    // fit) all their bases in ONE more: 16-byte chunks straight into LDS.  The per-iteration code then
    // never waits on global memory (it used to cost 3 dependent round trips per iteration).
    const uint32_t nrd = wave_first < n_reads ? (uint32_t)((n_reads - wave_first < FAST_READS_PER_WAVE) ? n_reads - wave_first : FAST_READS_PER_WAVE) : 0u;
    uint64_t myoff = 0;
    if ((uint32_t)lane <= nrd && nrd) myoff = offsets[wave_first + (uint32_t)lane];
    const uint64_t span_lo = __shfl(myoff, 0), span_hi = __shfl(myoff, (int)nrd);
    const uintptr_t raw_a0 = ((uintptr_t)bases + span_lo) & ~(uintptr_t)15;
    const uintptr_t raw_end = (uintptr_t)bases + span_hi;
    const bool bulk = nrd && (raw_end - raw_a0) <= (uintptr_t)(RAWB - 64);
    if (bulk) {
        // all chunks are requested before the first one is waited for: a chunk that is not wanted (past the wave's
        // reads) or not wholly inside the buffer re-reads the wave's first chunk instead of branching around the load
        const uintptr_t lim = (uintptr_t)bases + P.bases_bytes;
        constexpr int NCH = PAIR ? 5 : 3;
        uint4 v[NCH]; bool whole[NCH];
        const bool first_ok = raw_a0 + 16 <= lim;               // (false only for a buffer of < 16 bytes)
#pragma unroll
        for (int x = 0; x < NCH; x++) {
            const uintptr_t a = raw_a0 + 16u * (uint32_t)(lane + 64 * x);
            whole[x] = a < raw_end + 16 && a + 16 <= lim;
            v[x] = first_ok ? *(const uint4 *)(whole[x] ? a : raw_a0) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int x = 0; x < NCH; x++) {
            const uintptr_t a = raw_a0 + 16u * (uint32_t)(lane + 64 * x);
            if (whole[x]) *(uint4 *)(raw32 + 4 * (lane + 64 * x)) = v[x];
            else if (a < raw_end + 16)                          // the buffer ends inside this chunk
                for (int y = 0; y < 4; y++) raw32[4 * (lane + 64 * x) + y] = (a + 4 * y + 4 <= lim) ? *(const uint32_t *)(a + 4 * y) : 0u;
        }
    }
    wave_sync();

    // spectrum slot of the wave's first read: ONE 64-bit division per wave (it used to be a quarter of
    // all instructions when done per read); the following 15 reads step from it
    uint32_t slot0 = P.ring_base; uint64_t rem0 = 0;
    if (P.interval) {
        // a launch covers at most ring_n - 1 intervals (hulk_api.hip), so the quotient is found by a short
        // scalar loop; the two 64-bit divisions that stood here were ~3 % of the kernel's VALU instructions
        const uint64_t x = P.fill + wave_first;
        uint32_t xl = __builtin_amdgcn_readfirstlane((uint32_t)x), xh = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
        uint64_t xs = ((uint64_t)xh << 32) | xl;
        uint32_t t0 = 0;
        while (xs >= P.interval && t0 < P.ring_n) { xs -= P.interval; t0++; }
        if (xs >= P.interval) { t0 += (uint32_t)((xs / P.interval) % P.ring_n); xs %= P.interval; }   // not reached by libhulkhip's own launches
        rem0 = xs;
        slot0 = t0 + P.ring_base;
        while (slot0 >= P.ring_n) slot0 -= P.ring_n;
    }

    const bool crosses = P.interval && rem0 + FAST_READS_PER_WAVE > P.interval;
    // Per-read bookkeeping in 32 bits: offsets relative to the wave's first base.  (A wave whose 16 reads span
    // 2 GB or more hands all of them to the generic kernel, which works on the 64-bit offsets.)
    const bool wide = nrd && (span_hi - span_lo) >= 0x7fffffffull;
    const uint32_t myrel = (uint32_t)(myoff - span_lo);
    const uint32_t delta = (uint32_t)(((uintptr_t)bases + span_lo) - raw_a0);      // 0..15: first base within the staged bytes
    const bool lastwave = wave_first + nrd == n_reads;
    for (int it = 0; it < FAST_READS_PER_WAVE / RPI; it++) {
        if ((uint32_t)(RPI * it) >= nrd) break;
        const uint32_t idx = (uint32_t)(RPI * it + sub);       // read of the wave
        bool act = idx < nrd;                                  // group-uniform
        uint32_t hslot = slot0;
        if (crosses) {                                          // wave-uniform: an interval ends inside the wave's reads
            uint64_t x = rem0 + (uint64_t)idx;
            while (x >= P.interval) { x -= P.interval; hslot = hslot + 1 == P.ring_n ? 0u : hslot + 1; }
        }
        uint32_t a32 = 0; int32_t L = 0, npos = 0;
        {
            const uint32_t a = (uint32_t)__shfl((int)myrel, (int)idx), b = (uint32_t)__shfl((int)myrel, (int)idx + 1);
            if (act) { a32 = a; L = (int32_t)(b - a); }
        }
        bool defer = false;
        if (act) {
            if (wide) defer = true;
            else if (L < 1) { if (gl == 0) set_error(st, -3); act = false; }
            else if (L < w + k - 1) { if (gl == 0) set_error(st, -4); act = false; }
            else {
                npos = L - k + 1;
                if (npos > (PAIR ? 2 * 16 * w - (w - 1) : 16 * w) || L > (PAIR ? 512 : 256)) defer = true;
            }
        }
        // this group's part of the read: bases from posoff on, k-mer positions posoff .. posoff + 16w - 1
        const int32_t Lg = L - posoff;
        const int32_t nposg = npos - posoff < 0 ? 0 : (npos - posoff > 16 * w ? 16 * w : npos - posoff);
        if (dbg & 64u) { sink += a32 + (uint32_t)npos + hslot; continue; }   // ablation: per-iteration bookkeeping only
        // ---- stage 16 bases per lane: ASCII -> 2-bit pack (one dword per lane), detect code 4
        bool sawN = false;
        if (act && !defer) {
            const int32_t p = 16 * gl;
            uint32_t pack = 0;
            if (p < Lg) {
                const uint32_t ro = a32 + (uint32_t)posoff + (uint32_t)p;     // first of the lane's bytes, from the wave's first base
                unsigned sh;
                uint32_t d[5];
                if (bulk) {
                    const uint32_t lo = ro + delta;                // raw_a0 is 16-byte aligned
                    const uint32_t *src = raw32 + (lo >> 2);
                    sh = (lo & 3u) * 8u;
#pragma unroll
                    for (int x = 0; x < 5; x++) d[x] = src[x];
                } else {
                    const uintptr_t addr = (uintptr_t)bases + span_lo + ro;
                    const uintptr_t al = addr & ~(uintptr_t)3, end = (uintptr_t)bases + P.bases_bytes;
                    sh = (unsigned)(addr & 3) * 8;
#pragma unroll
                    for (int x = 0; x < 5; x++) d[x] = (al + 4 * (x + 1) <= end) ? *(const uint32_t *)(al + 4 * x) : 0u;
                }
                // ASCII -> 2-bit, four bases per dword, no table: fold the case, code = (c >> 1 ^ c >> 2) & 3
                // (A 0, C 1, G 2, T 3), and prove it by mapping the codes back to letters with one v_perm_b32:
                // any byte that does not come back (N, U, 0..3, anything else) defers the read to the generic
                // kernel, which has the full nt4 table.  (c * 0x01041040) >> 24 gathers the four codes of a dword
                // into one byte.  Bytes past the read's end are the next read's (or zero): they only reach k-mer
                // positions that are not reported, and at worst defer a read that did not need it.
                uint32_t bad = 0;
#pragma unroll
                for (int x = 0; x < 4; x++) {
                    const uint32_t by = __builtin_amdgcn_alignbit(d[x + 1], d[x], sh);   // (sh = 0: d[x])
                    const uint32_t up = by & 0xDFDFDFDFu;
                    const uint32_t e = up >> 1;
                    const uint32_t c = (e ^ (e >> 1)) & 0x03030303u;
                    bad |= __builtin_amdgcn_perm(0u, 0x54474341u, c) ^ up;
                    pack |= ((c * 0x01041040u) >> 24) << (8 * x);
                }
                if (Lg - p < 16 && lastwave && idx + 1 == nrd) {
                    // the last read of the call: what follows it is not a read; test its own bytes only
                    const int nv = (int)(Lg - p);
                    bad = 0;
#pragma unroll
                    for (int x = 0; x < 4; x++) {
                        const uint32_t by = __builtin_amdgcn_alignbit(d[x + 1], d[x], sh);   // (sh = 0: d[x])
                        const uint32_t up = by & 0xDFDFDFDFu, e = up >> 1, c = (e ^ (e >> 1)) & 0x03030303u;
                        const int nb = nv - 4 * x;
                        const uint32_t m = nb >= 4 ? ~0u : nb <= 0 ? 0u : (1u << (8 * nb)) - 1u;
                        bad |= (__builtin_amdgcn_perm(0u, 0x54474341u, c) ^ up) & m;
                    }
                }
                sawN = bad != 0;
            }
            pk32[gl] = pack;
            if (gl < 4) pk32[16 + gl] = 0;                     // slack for 3-dword window reads
        }
        {
            // code 4 anywhere in the read defers it as a whole (both groups of a pair must agree)
            const uint32_t gN = PAIR ? (uint32_t)(__ballot(sawN) >> (lane & 32)) : (uint32_t)(__ballot(sawN) >> gsh) & 0xffffu;
            if (gN) defer = true;
        }
        wave_sync();
        if (dbg & 32u) { sink += pk32[gl]; wave_sync(); continue; }      // ablation: staging only

        // ---- phase A: rolling k-mers over the own block (registers)
        const int32_t p0 = gl * w;                              // first position of the lane's block, within the group
        const int32_t ap0 = posoff + p0;                        // ... within the read
        const bool mine = act && !defer && p0 < nposg;
        uint32_t validbits = 0;
        uint64_t X[WM];
#pragma unroll
        for (int t = 0; t < WM; t++) X[t] = XN;
        if (mine) {
            // Positions past the read's end are always the END of a lane's block and of the read: their values only
            // reach windows that are not reported (validbits gates every report), so they are computed like any
            // other instead of being replaced by "no value" one by one.
            validbits = (1u << (nposg - p0 < w ? nposg - p0 : w)) - 1u;
            const int32_t span0 = ap0 + k - 1 - w + 2;
            uint64_t f = 0, r = 0;
            uint32_t nb = 0;                                    // next <=15 bases, 2 bits each
            uint64_t Bb = 0, Cl = 0;
            if (DX) {
                // the block's w+k-1 bases fit one 64-bit window: every k-mer of the block is a shift + mask of the
                // window in the forward (first base on top: Bb) or the complemented (first base at the bottom: Cl)
                // layout, instead of the rolling update per base
                constexpr int NBW = KC + WM - 1;
                const uint32_t bo = 2u * (uint32_t)p0, d = bo >> 5, o = bo & 31u;
                const uint64_t lo = (uint64_t)pk32[d] | ((uint64_t)pk32[d + 1] << 32);
                uint64_t Wl = o ? (lo >> o) | ((uint64_t)pk32[d + 2] << (64 - o)) : lo;
                if (NBW < 32) Wl &= (1ull << (2 * NBW)) - 1;
                const uint64_t rv = __brevll(Wl) >> (64 - 2 * NBW);
                Bb = ((rv >> 1) & 0x5555555555555555ull) | ((rv & 0x5555555555555555ull) << 1);
                Cl = ~Wl;
            } else {
                {
                    const uint32_t bo = 2u * (uint32_t)p0, d = bo >> 5, o = bo & 31u;
                    const uint64_t lo = (uint64_t)pk32[d] | ((uint64_t)pk32[d + 1] << 32);
                    uint64_t W = o ? (lo >> o) | ((uint64_t)pk32[d + 2] << (64 - o)) : lo;
                    W &= mask;
                    const uint64_t rev = __brevll(W) >> (64 - 2 * k);
                    f = ((rev >> 1) & 0x5555555555555555ull) | ((rev & 0x5555555555555555ull) << 1);
                    r = (~W) & mask;
                }
                {
                    const uint32_t bo = 2u * (uint32_t)(p0 + k), d = bo >> 5, o = bo & 31u;
                    const uint64_t lo = (uint64_t)pk32[d] | ((uint64_t)pk32[d + 1] << 32);
                    nb = (uint32_t)(lo >> o);
                }
            }
#pragma unroll
            for (int t = 0; t < WM; t++) {
                if (DX) {
                    // dword form (v_alignbit + v_bfe) of f = (Bb >> 2(WM-1-t)) & mask, r = (Cl >> 2t) & mask
                    constexpr int HB = 2 * KC - 32;
                    const int sf = 2 * (WM - 1 - t), sr = 2 * t;
                    const uint32_t bl = (uint32_t)Bb, cl = (uint32_t)Cl;
                    uint32_t bh = (uint32_t)(Bb >> 32), ch = (uint32_t)(Cl >> 32);
                    asm("" : "+v"(bh), "+v"(ch));   // opaque: or hipcc re-fuses the dwords into 64-bit shifts + masks
                    const uint32_t fl = sf ? __builtin_amdgcn_alignbit(bh, bl, sf) : bl, fh = __builtin_amdgcn_ubfe(bh, sf, HB);
                    const uint32_t rl = sr ? __builtin_amdgcn_alignbit(ch, cl, sr) : cl, rh = __builtin_amdgcn_ubfe(ch, sr, HB);
                    f = ((uint64_t)fh << 32) | fl;
                    r = ((uint64_t)rh << 32) | rl;
                } else if (t) {
                    const uint64_t c = nb & 3u; nb >>= 2;
                    f = (f << 2 | c) & mask;
                    r = (r >> 2) | ((3ull ^ c) << shift);
                }
                if (t < w) {
                    const uint64_t canon = umin64<FM>(f, r);
                    int32_t span = span0 + t;
                    if (span >= k) span = k;
                    uint64_t x;
                    if (HP && !(dbg & 16u)) x = hash64_pack_kc<HP ? KC : 21>(canon, (uint32_t)span);
                    else x = ((dbg & 16u) ? canon * 0x9E3779B97F4A7C15ull : hash64_fm<FM>(canon, mask)) << 8 | (uint64_t)(int64_t)span;
                    // f == r (a k-mer that is its own reverse complement: even k only, reads with N never get
                    // here) is skipped by the reference: the position neither reports nor takes part in a window
                    if (!(k & 1) && f == r) { x = XN; validbits &= ~(1u << t); }
                    X[t] = x;
                }
            }
        }
        if (dbg & 8u) {
#pragma unroll
            for (int t = 0; t < WM; t++) sink += (uint32_t)X[t];
            wave_sync();
            continue;
        }

        // ---- phase B: windowed minimum m(pos) = min(prev block's suffix-min, own prefix-min)
        uint32_t startbits = 0;
        {
            // own suffix minima h[t] = min(X[t..w-1]); the next lane needs h[t+1] as hp[t]
            uint64_t hp[WM];
            {
                uint64_t h = XN;
#pragma unroll
                for (int t = WM - 1; t >= 0; t--) {
                    if (t < w) h = umin64<FM>(X[t], h);
                    hp[t] = h;                                 // h[t]
                }
            }
            const uint64_t whole = dpp_row_shr1_u64(hp[0]);    // min of the whole previous block
            // lane 0 of a row has no previous block: DPP hands it zeros, which "no value" (+inf, FM) differs from in
            // the upper dword only — one v_or per value instead of a 64-bit select
            const uint64_t xnfix = gl == 0 ? XN : 0ull;
#pragma unroll
            for (int t = 0; t < WM - 1; t++) {
                const uint64_t v = dpp_row_shr1_u64(hp[t + 1]);
                hp[t] = FM ? (v | (xnfix & 0xffffffff00000000ull)) : (v | xnfix);
            }
            hp[WM - 1] = XN;
            const uint32_t pv = dpp_row_shr1(validbits);
            // m(pos) for the block, and for each position whether it continues the previous position's value
            uint32_t eqbits = 0;
            uint64_t g = XN, pm = whole;
#pragma unroll
            for (int t = 0; t < WM; t++) {
                g = umin64<FM>(X[t], g);
                const uint64_t hpt = (t + 1 < w) ? hp[t] : XN;
                const uint64_t m = umin64<FM>(hpt, g);
                if (t < w) { eqbits |= (m == pm) ? (1u << t) : 0u; pm = m; }
                X[t] = m;
            }
            // positions that report: valid, at or past the first window end (i >= w-1: always when k >= w), and not
            // one of the w-1 context positions of a pair's second group
            uint32_t emitbits = validbits;
            {
                const int32_t t1 = (w - 1) - (k - 1) - ap0;
                if (t1 > 0) emitbits &= ~((1u << t1) - 1u);
                if (PAIR && half) { const int32_t t2 = (w - 1) - p0; if (t2 > 0) emitbits &= ~((1u << t2) - 1u); }
            }
            const uint32_t pe0 = (gl > 0 && ((pv >> (w - 1)) & 1u) && (ap0 - 1 + k - 1 >= w - 1) && !(half && p0 - 1 < w - 1)) ? 1u : 0u;
            // a run starts where a reporting position does not repeat the value of a reporting predecessor
            startbits = emitbits & ~(((emitbits << 1) | pe0) & eqbits);
        }
        // ---- compact the run-start values of the read into the group's candidate list (LDS)
        uint32_t total;
        {
            const uint32_t cnt = (uint32_t)__popc(startbits);
            uint32_t incl = cnt;
            incl += dpp_row_shr1(incl);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, true);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, true);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, true);
            total = (uint32_t)__shfl((int)incl, (lane & 48) | 15);
            bool ovf = total > (uint32_t)FAST_CAND;                         // very repetitive read: generic kernel
            if (PAIR) ovf = ((uint32_t)(__ballot(ovf) >> (lane & 32))) != 0u;  // ... for both halves of it
            if (ovf) { defer = true; total = 0; }
            else {
                uint32_t at = incl - cnt;
#pragma unroll
                for (int t = 0; t < WM; t++)
                    if ((startbits >> t) & 1u) cs[at++] = X[t];
            }
        }
        if (act && defer) {
            if (gl == 0 && half == 0) { const uint32_t at = atomicAdd(slow_count, 1u); slow_list[at] = (uint32_t)(wave_first + idx); }
            act = false;
        }
        wave_sync();

        // ---- per-read set + list append: one candidate per lane per round (<= 4 rounds).  A
        // candidate is new iff its compare-and-swap finds the slot empty; new values go straight to
        // the wave's region of the minimizer list (consecutive ranks = consecutive addresses).
        uint32_t myslot[FAST_CAND / 16];
        uint32_t newmask = 0;
#pragma unroll
        for (int rnd = 0; rnd < FAST_CAND / 16; rnd++) {
            const uint32_t c = (uint32_t)gl + 16u * (uint32_t)rnd;
            myslot[rnd] = 0;
            if (!__any((int)(c < total))) break;
            bool isnew = false; uint64_t x = 0;
            if (c < total) {
                x = cs[c];
                uint32_t sl = ((uint32_t)(x >> 8) ^ (uint32_t)(x >> 37)) & (FAST_TAB - 1);
                unsigned long long o = (dbg & 4u) ? (unsigned long long)TAB_EMPTY
                                                  : atomicCAS((unsigned long long *)&tab[sl], (unsigned long long)TAB_EMPTY, (unsigned long long)x);
                while (o != TAB_EMPTY && o != x) {                 // occupied by another value: probe on
                    sl = (sl + 1) & (FAST_TAB - 1);
                    o = atomicCAS((unsigned long long *)&tab[sl], (unsigned long long)TAB_EMPTY, (unsigned long long)x);
                }
                isnew = (o == TAB_EMPTY);
                myslot[rnd] = sl;
            }
            const uint64_t nbal = __ballot(isnew);
            if (nbal) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(nbal >> 32),
                                          __builtin_amdgcn_mbcnt_lo((uint32_t)nbal, 0u));
                if (isnew) {
                    newmask |= 1u << rnd;
                    if (dbg & 1u) sink += (uint32_t)x; else { xl[wcount + rank] = x; sl8[wcount + rank] = (uint8_t)hslot; }
                }
                wcount += (uint32_t)__popcll(nbal);
            }
        }
        wave_sync();
        // empty the set again: only the slots this lane filled
#pragma unroll
        for (int rnd = 0; rnd < FAST_CAND / 16; rnd++)
            if ((newmask >> rnd) & 1u) tab[myslot[rnd]] = TAB_EMPTY;
        wave_sync();
    }
    if (lane == 0 && wave_first < n_reads) ml.cnt[region] = wcount;
    if (dbg && sink == 0xdeadbeefu) xl[0] = sink;
    __shared__ unsigned long long blk_nmin[4];
    if (lane == 0) blk_nmin[wid] = wcount;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = blk_nmin[0] + blk_nmin[1] + blk_nmin[2] + blk_nmin[3];
        if (t) atomicAdd(&min_slots[blockIdx.x & (MIN_SLOTS - 1)], t);   // boss.minimizerCounter, spread over slots
    }
}

// ------------------------------------------------------------------------------------------
// K1b: jump hash of the minimizer list.  One wave per region; lanes take the region's values
// round-robin (lane l: l, l+64, ...) with the next value prefetched, so every lane stays busy
// with its own chain of ~ln(k^4) fp64 steps (no lock-step tail per 64 values); 8 waves/SIMD.
// Output: key = spectrum slot << 20 | bin  (k^4 < 2^20 for k <= 31).
// ------------------------------------------------------------------------------------------
// The step loop of k_jump_bin in assembly (fixed registers v40..v57, s60..s63): hipcc's version of the same loop
// carries two v_mov_b64 and a dozen scalar mask instructions per pair of steps; this one is the 17 VALU
// instructions of a step plus v_cmp / s_and and the exit test, the LCG state ping-ponging between v[40:41] and
// v[42:43].  Lanes that reach p >= n leave the exec mask and keep their t.  The loop ends when at most `cut`
// lanes are still running (cut = 0: when none is): chains take 12.8 +- 3.5 steps, so the last few lanes of a round
// of 64 would keep the whole wave busy for ~24 — they are handed over instead (`left` = their mask, key/t = their
// state at a step boundary) and finished by k_jump_left in a denser wave.
// p >= n is tested on the upper dwords alone: p is a non-negative finite double and n < 2^20 is an integer whose
// double has a zero lower dword, so bits(p) >= bits(n) <=> hi(p) >= hi(n)  (v_cmp_lt_u32 instead of v_cmp_nge_f64).
__device__ __forceinline__ double jump_steps_asm(uint32_t &klo, uint32_t &khi, double fn, double t0, uint32_t cut,
                                                 unsigned long long &left) {
    uint32_t tlo, thi, mlo, mhi, olo, ohi;
    const uint64_t fb = (uint64_t)__double_as_longlong(fn), tb = (uint64_t)__double_as_longlong(t0);
    const uint32_t flo = (uint32_t)fb, fhi = (uint32_t)(fb >> 32), t0lo = (uint32_t)tb, t0hi = (uint32_t)(tb >> 32);
#define HULK_JSTEP(KS_LO, KS_HI, KD, KD_HI, EXIT)                                    \
    "v_mad_u64_u32 " KD ", s[62:63], " KS_LO ", %[alo], 1\n\t"                       \
    "v_mul_lo_u32 v54, " KS_LO ", %[ahi]\n\t"                                        \
    "v_mul_lo_u32 v55, " KS_HI ", %[alo]\n\t"                                        \
    "v_add3_u32 " KD_HI ", v55, " KD_HI ", v54\n\t"                                  \
    "v_lshrrev_b32 v54, 1, " KD_HI "\n\t"                                            \
    "v_add_u32 v54, 1, v54\n\t"                                                      \
    "v_cvt_f64_u32 v[46:47], v54\n\t"                                                \
    "v_add_u32 v47, 0xfe100000, v47\n\t"                                             \
    "v_rcp_f64 v[48:49], v[46:47]\n\t"                                               \
    "s_nop 0\n\t"                                                                    \
    "v_fma_f64 v[50:51], -v[46:47], v[48:49], 1.0\n\t"                               \
    "v_fma_f64 v[48:49], v[48:49], v[50:51], v[48:49]\n\t"                           \
    "v_fma_f64 v[50:51], -v[46:47], v[48:49], 1.0\n\t"                               \
    "v_fma_f64 v[48:49], v[48:49], v[50:51], v[48:49]\n\t"                           \
    "v_fma_f64 v[52:53], v[44:45], v[48:49], v[48:49]\n\t"                           \
    "v_cmp_lt_u32 vcc, v53, v57\n\t"                                                 \
    "s_and_b64 exec, exec, vcc\n\t"                                                  \
    "v_trunc_f64 v[44:45], v[52:53]\n\t"                                             \
    "s_bcnt1_i32_b64 s62, exec\n\t"                                                  \
    "s_cmp_le_u32 s62, %[cut]\n\t"                                                   \
    "s_cbranch_scc1 " EXIT "\n\t"
    asm volatile(
        "s_mov_b64 s[60:61], exec\n\t"
        "v_mov_b32 v40, %[klo]\n\t"
        "v_mov_b32 v41, %[khi]\n\t"
        "v_mov_b32 v56, %[flo]\n\t"
        "v_mov_b32 v57, %[fhi]\n\t"
        "v_mov_b32 v44, %[t0lo]\n\t"
        "v_mov_b32 v45, %[t0hi]\n\t"
        "1:\n\t"
        HULK_JSTEP("v40", "v41", "v[42:43]", "v43", "3f")
        HULK_JSTEP("v42", "v43", "v[40:41]", "v41", "2f")
        "s_branch 1b\n\t"
        "3:\n\t"                                   // left after the first half: the live key is in v[42:43]
        "v_mov_b32 v40, v42\n\t"
        "v_mov_b32 v41, v43\n\t"
        "2:\n\t"
        "s_mov_b32 %[mlo], exec_lo\n\t"
        "s_mov_b32 %[mhi], exec_hi\n\t"
        "s_mov_b64 exec, s[60:61]\n\t"
        "v_mov_b32 %[tlo], v44\n\t"
        "v_mov_b32 %[thi], v45\n\t"
        "v_mov_b32 %[olo], v40\n\t"
        "v_mov_b32 %[ohi], v41\n\t"
        : [tlo] "=v"(tlo), [thi] "=v"(thi), [olo] "=v"(olo), [ohi] "=v"(ohi), [mlo] "=s"(mlo), [mhi] "=s"(mhi)
        : [klo] "v"(klo), [khi] "v"(khi), [flo] "v"(flo), [fhi] "v"(fhi), [t0lo] "v"(t0lo), [t0hi] "v"(t0hi),
          [alo] "s"(0x87B0B0FDu), [ahi] "s"(0x27BB2EE6u), [cut] "s"(cut)
        : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55",
          "v56", "v57", "s60", "s61", "s62", "s63", "vcc", "scc", "memory");
#undef HULK_JSTEP
    klo = olo; khi = ohi;
    left = ((unsigned long long)mhi << 32) | mlo;
    return __longlong_as_double((long long)(((uint64_t)thi << 32) | tlo));
}

__global__ __launch_bounds__(256) void k_jump_bin(MinimizerList ml, uint32_t n_regions, int32_t num_bins, int use_c,
                                                  uint32_t cut) {
    const int lane = lane_id();
    const uint32_t region = (uint32_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (region >= n_regions) return;
    const uint32_t cnt = ml.cnt[region];
    const uint64_t *xl = ml.x + (size_t)region * ml.rcap;
    const uint8_t *sl = ml.slot + (size_t)region * ml.rcap;
    uint32_t *kl = ml.key + ml.off[region];                   // dense: regions back to back
    uint4 *lo = ml.lo + (size_t)region * JUMP_LO_CAP;
    uint32_t nleft = 0;                                        // wave-uniform: chains handed to k_jump_left so far
    uint32_t idx = (uint32_t)lane;
    uint64_t nx = 0; uint32_t ns = 0;
    if (idx < cnt) { nx = xl[idx]; ns = sl[idx]; }
    const double fn = (double)num_bins;
    while (idx < cnt) {
        uint64_t key = nx; const uint32_t slot = ns;
        const uint32_t nidx = idx + 64;
        if (nidx < cnt) { nx = xl[nidx]; ns = sl[nidx]; }      // prefetch the lane's next value
        // Literally the reference's step: j = int64(float64(b+1) * (float64(1<<31) / float64(r))).  float64(b) = t is
        // carried; (t + 1) * q is ONE fma(t, q, q) — the exact product rounded once, as the multiplication is —
        // so a step needs no add and no ldexp.
        double t = 0.0;                                         // float64(b), b = 0 before the first step
        if (!use_c) {
            uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
            unsigned long long left = 0;
            t = jump_steps_asm(klo, khi, fn, 0.0, cut, left);
            const bool mine = (left >> lane) & 1ull;            // this lane's chain is not finished
            if (left) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(left >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)left, 0u));
                const uint32_t pos = nleft + rank;
                bool stored = false;
                if (mine && pos < (uint32_t)JUMP_LO_CAP) {
                    lo[pos] = make_uint4(klo, khi, (uint32_t)(int32_t)t, idx | (slot << 16));
                    stored = true;
                }
                const unsigned long long spill = __ballot(mine && !stored);
                if (spill) {                                    // the region's hand-over area is full: finish here
                    if (mine && !stored) { unsigned long long none; t = jump_steps_asm(klo, khi, fn, t, 0u, none); }
                }
                nleft += (uint32_t)__popcll(left);
                if (mine && stored) { idx = nidx; continue; }
            }
        } else
        for (;;) {
            key = key * 2862933555777941757ull + 1;
            double q = quot31_exact((uint32_t)(key >> 33) + 1u);
            double p = __builtin_fma(t, q, q);
            if (p >= fn) break;                                 // j >= n: t is the bucket
            t = __builtin_trunc(p);                             // j = int64(p): exact, < 2^31
            key = key * 2862933555777941757ull + 1;
            q = quot31_exact((uint32_t)(key >> 33) + 1u);
            p = __builtin_fma(t, q, q);
            if (p >= fn) break;
            t = __builtin_trunc(p);
        }
        const int32_t res = (int32_t)t;
        kl[idx] = (slot << 20) | (uint32_t)res;
        idx = nidx;
    }
    if (lane == 0) ml.lo_cnt[region] = nleft < (uint32_t)JUMP_LO_CAP ? nleft : (uint32_t)JUMP_LO_CAP;
}

// finishes the chains k_jump_bin handed over: one wave per region, at most one round
__global__ __launch_bounds__(256) void k_jump_left(MinimizerList ml, uint32_t n_regions, int32_t num_bins) {
    const int lane = lane_id();
    const uint32_t region = (uint32_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (region >= n_regions) return;
    const uint32_t n = ml.lo_cnt[region];
    if ((uint32_t)lane >= n) return;
    const uint4 st = ml.lo[(size_t)region * JUMP_LO_CAP + lane];
    uint32_t klo = st.x, khi = st.y;
    unsigned long long none;
    const double t = jump_steps_asm(klo, khi, (double)num_bins, (double)(int32_t)st.z, 0u, none);
    ml.key[ml.off[region] + (st.w & 0xffffu)] = ((st.w >> 16) << 20) | (uint32_t)(int32_t)t;
}

// ------------------------------------------------------------------------------------------
// K1c: k-mer spectrum from the key list WITHOUT global atomics.  Random global atomicAdd runs at
// ~27 G lane-ops/s on this chip whatever the scope or footprint (tools/ubench/atomics*.hip) — 1 ms
// per 10^6 reads here — while LDS atomics and coalesced traffic are an order of magnitude cheaper:
// workgroup (r, t) owns bins [r*RANGE, (r+1)*RANGE) of spectrum t, counts them in LDS over the
// (L2/Infinity-Cache resident) keys of that interval's reads and adds the range to the spectrum
// with plain coalesced read-modify-writes — it is the only writer of those bins in this launch.
// ------------------------------------------------------------------------------------------
constexpr int HIST_RANGE = 32768;     // bins per workgroup (128 KB of LDS)

// exclusive prefix sum of the region counts: per-block sums, then one block per 1024 regions
__global__ __launch_bounds__(1024) void k_region_bsum(const uint32_t *__restrict__ cnt, uint32_t *__restrict__ bsum,
                                                      uint32_t n_regions) {
    __shared__ uint32_t wsum[16];
    const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
    uint32_t v = i < n_regions ? cnt[i] : 0u;
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t t = 0; for (int x = 0; x < 16; x++) t += wsum[x]; bsum[blockIdx.x] = t; }
}
__global__ __launch_bounds__(1024) void k_region_offsets(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ bsum,
                                                         uint32_t *__restrict__ off, uint32_t n_regions,
                                                         uint32_t *__restrict__ nib_over, uint32_t *__restrict__ zero_word) {
    if (nib_over && blockIdx.x == 0 && threadIdx.x < RING_MAX) nib_over[threadIdx.x] = 0;
    if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0;      // the next launch's slow-list counter
    __shared__ uint32_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t i = blockIdx.x * 1024u + (uint32_t)tid;
    uint32_t before = 0;
    for (uint32_t b = 0; b < blockIdx.x; b++) before += bsum[b];       // same address for the whole block: broadcast
    const uint32_t v = i < n_regions ? cnt[i] : 0u;
    const uint32_t incl = wave_scan_incl(v);
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    for (int x = 0; x < wid; x++) before += wsum[x];
    if (i < n_regions) off[i] = before + incl - v;
    if (i + 1 == n_regions) off[n_regions] = before + incl;
}

// workgroup (r, t, part): LDS spectrum of bins [r*RANGE, (r+1)*RANGE) over part `part` of interval t's keys
__global__ __launch_bounds__(1024) void k_range_hist(MinimizerList ml, uint32_t n_regions,
                                                     uint32_t *__restrict__ partial, MinimizerParams P,
                                                     uint32_t n_spectra, uint32_t n_parts, uint64_t n_reads, int nranges,
                                                     const uint32_t *__restrict__ only_if, uint32_t *__restrict__ hists_direct) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *lh = (uint32_t *)smem;
    // XCD-aware order (workgroup b lands on XCD b % 8): the nranges workgroups that stream the SAME keys
    // (one (spectrum, part) pair, different bin ranges) get consecutive slots of ONE XCD, so its L2 serves
    // all but the first of them — otherwise every range re-fetches the keys over the fabric (nranges x)
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int pr = (seq / nranges) * 8 + xcd, r = seq % nranges;
    if (pr >= (int)(n_spectra * n_parts)) return;
    const int t = pr % (int)n_spectra, part = pr / (int)n_spectra;
    if (only_if && !only_if[t]) return;                          // fallback mode: only spectra whose nibble count overflowed
    const int tid = threadIdx.x;
    for (int i = tid; i < HIST_RANGE; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    // reads of spectrum t: local read index rd with (fill + rd) / interval == t   (all reads if interval == 0)
    uint64_t rd0 = 0, rd1 = n_reads;
    if (P.interval) {
        const uint64_t lo = (uint64_t)t * P.interval, hi = lo + P.interval;
        rd0 = lo > P.fill ? lo - P.fill : 0;
        rd1 = hi > P.fill ? hi - P.fill : 0;
        if (rd1 > n_reads) rd1 = n_reads;
    }
    const uint32_t slot = P.interval ? (uint32_t)(((uint64_t)t + P.ring_base) % P.ring_n) : P.ring_base;
    if (rd0 < rd1) {
        const uint32_t g0 = (uint32_t)(rd0 / FAST_READS_PER_WAVE), g1 = (uint32_t)((rd1 - 1) / FAST_READS_PER_WAVE);
        const uint32_t a = ml.off[g0], b = ml.off[(g1 + 1 < n_regions ? g1 + 1 : n_regions)];
        const uint32_t len = b - a, per = (len + n_parts - 1) / n_parts;
        const uint32_t lo = a + (uint32_t)part * per, hi = (lo + per < b) ? lo + per : b;
        const uint32_t want = (slot << 5) | (uint32_t)r;           // key >> 15
        const uint32_t *kl = ml.key;
        // head up to 16-byte alignment, then 4 keys per lane per load with 4 loads in flight, then the tail
        uint32_t i = lo;
        const uint32_t head_end = ((lo + 3u) & ~3u) < hi ? ((lo + 3u) & ~3u) : hi;
        if (i + (uint32_t)tid < head_end) { const uint32_t k = kl[i + tid]; if ((k >> 15) == want) atomicAdd(&lh[k & (HIST_RANGE - 1)], 1u); }
        i = head_end;
        const uint4 *k4 = (const uint4 *)(kl + i);
        const uint32_t n4 = (hi - i) / 4u;
        uint32_t j = (uint32_t)tid;
#define HULK_COUNT4(q)                                                                         \
        { if ((q.x >> 15) == want) atomicAdd(&lh[q.x & (HIST_RANGE - 1)], 1u);                 \
          if ((q.y >> 15) == want) atomicAdd(&lh[q.y & (HIST_RANGE - 1)], 1u);                 \
          if ((q.z >> 15) == want) atomicAdd(&lh[q.z & (HIST_RANGE - 1)], 1u);                 \
          if ((q.w >> 15) == want) atomicAdd(&lh[q.w & (HIST_RANGE - 1)], 1u); }
        for (; j + 3u * 1024u < n4; j += 4u * 1024u) {
            const uint4 q0 = k4[j], q1 = k4[j + 1024u], q2 = k4[j + 2048u], q3 = k4[j + 3072u];
            HULK_COUNT4(q0) HULK_COUNT4(q1) HULK_COUNT4(q2) HULK_COUNT4(q3)
        }
        for (; j < n4; j += 1024u) { const uint4 q0 = k4[j]; HULK_COUNT4(q0) }
#undef HULK_COUNT4
        const uint32_t tail = i + n4 * 4u + (uint32_t)tid;
        if (tail < hi) { const uint32_t k = kl[tail]; if ((k >> 15) == want) atomicAdd(&lh[k & (HIST_RANGE - 1)], 1u); }
    }
    __syncthreads();
    const int32_t nb = P.num_bins - r * HIST_RANGE;
    if (hists_direct) {                                          // recount mode: straight into the spectrum (coalesced atomics)
        uint32_t *h = hists_direct + (size_t)slot * (size_t)P.num_bins + (size_t)r * HIST_RANGE;
        for (int i = tid; i < HIST_RANGE && i < nb; i += blockDim.x) if (lh[i]) atomicAdd(&h[i], lh[i]);
        return;
    }
    uint32_t *out = partial + ((size_t)part * n_spectra + t) * (size_t)P.num_bins + (size_t)r * HIST_RANGE;
    for (int i = tid; i < HIST_RANGE && i < nb; i += blockDim.x) out[i] = lh[i];
}

// spectrum[slot_t][bin] += sum over parts    grid = (blocks, n_spectra)
// K1c': the same spectrum with FOUR-BIT counters, so that one workgroup holds a whole range of 2^18 bins in LDS
// (all 194,481 bins at k = 21) and every key is read ONCE instead of once per 32768-bin range (k_range_hist was
// bound by those re-reads through L2).  A part is ~131 k keys over ~2*10^5 bins, so a counter reaching 16 needs
// grossly repetitive input (Poisson mean < 1 per bin); it cannot go unnoticed: every ds_add returns the previous word, a previous nibble
// of 15 raises nib_over[t] and k_range_hist / k_merge_hist recount that spectrum exactly (they return at once
// otherwise).  Layout of a part: words of 8 nibbles, bin b -> word b >> 3, nibble b & 7.
constexpr int NIB_BINS = 262144;                       // bins per range (128 KB of LDS)
constexpr int NIB_WORDS = NIB_BINS / 8;
__global__ __launch_bounds__(1024) void k_nibble_hist(MinimizerList ml, uint32_t n_regions, MinimizerParams P,
                                                      uint32_t n_spectra, uint32_t n_parts, uint64_t n_reads, int nranges) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *lw = (uint32_t *)smem;
    // XCD-aware order as in k_range_hist: the ranges of one (spectrum, part) pair share an XCD's L2
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int pr = (seq / nranges) * 8 + xcd, r = seq % nranges;
    if (pr >= (int)(n_spectra * n_parts)) return;
    const int t = pr % (int)n_spectra, part = pr / (int)n_spectra;
    const int tid = threadIdx.x;
    const int32_t rbase = r * NIB_BINS;
    const int32_t rbins = P.num_bins - rbase < NIB_BINS ? P.num_bins - rbase : NIB_BINS;
    const int words = (rbins + 7) >> 3;
    for (int i = tid; i < words; i += blockDim.x) lw[i] = 0;
    __syncthreads();
    uint64_t rd0 = 0, rd1 = n_reads;
    if (P.interval) {
        const uint64_t lo = (uint64_t)t * P.interval, hi = lo + P.interval;
        rd0 = lo > P.fill ? lo - P.fill : 0;
        rd1 = hi > P.fill ? hi - P.fill : 0;
        if (rd1 > n_reads) rd1 = n_reads;
    }
    const uint32_t slot = P.interval ? (uint32_t)(((uint64_t)t + P.ring_base) % P.ring_n) : P.ring_base;
    bool over = false;
    if (rd0 < rd1) {
        const uint32_t g0 = (uint32_t)(rd0 / FAST_READS_PER_WAVE), g1 = (uint32_t)((rd1 - 1) / FAST_READS_PER_WAVE);
        const uint32_t a = ml.off[g0], b = ml.off[(g1 + 1 < n_regions ? g1 + 1 : n_regions)];
        const uint32_t len = b - a, per = (len + n_parts - 1) / n_parts;
        const uint32_t lo = a + (uint32_t)part * per, hi = (lo + per < b) ? lo + per : b;
        const uint32_t *kl = ml.key;
#define HULK_NIB1(k)                                                                                   \
        if (((k) >> 20) == slot) {                                                                     \
            const uint32_t rel = ((k) & 0xFFFFFu) - (uint32_t)rbase;                                   \
            if (rel < (uint32_t)rbins) {                                                               \
                const uint32_t sh = (rel & 7u) * 4u;                                                   \
                const uint32_t old = atomicAdd(&lw[rel >> 3], 1u << sh);                               \
                over |= ((old >> sh) & 15u) == 15u;                                                    \
            }                                                                                          \
        }
        uint32_t i = lo;
        const uint32_t head_end = ((lo + 3u) & ~3u) < hi ? ((lo + 3u) & ~3u) : hi;
        if (i + (uint32_t)tid < head_end) { const uint32_t k = kl[i + tid]; HULK_NIB1(k) }
        i = head_end;
        const uint4 *k4 = (const uint4 *)(kl + i);
        const uint32_t n4 = hi > i ? (hi - i) / 4u : 0u;
        uint32_t j = (uint32_t)tid;
        for (; j + 3u * 1024u < n4; j += 4u * 1024u) {
            const uint4 q0 = k4[j], q1 = k4[j + 1024u], q2 = k4[j + 2048u], q3 = k4[j + 3072u];
            HULK_NIB1(q0.x) HULK_NIB1(q0.y) HULK_NIB1(q0.z) HULK_NIB1(q0.w)
            HULK_NIB1(q1.x) HULK_NIB1(q1.y) HULK_NIB1(q1.z) HULK_NIB1(q1.w)
            HULK_NIB1(q2.x) HULK_NIB1(q2.y) HULK_NIB1(q2.z) HULK_NIB1(q2.w)
            HULK_NIB1(q3.x) HULK_NIB1(q3.y) HULK_NIB1(q3.z) HULK_NIB1(q3.w)
        }
        for (; j < n4; j += 1024u) { const uint4 q0 = k4[j]; HULK_NIB1(q0.x) HULK_NIB1(q0.y) HULK_NIB1(q0.z) HULK_NIB1(q0.w) }
        const uint32_t tail = i + n4 * 4u + (uint32_t)tid;
        if (tail < hi) { const uint32_t k = kl[tail]; HULK_NIB1(k) }
#undef HULK_NIB1
    }
    if (__any((int)over) && (tid & 63) == 0) ml.nib_over[t] = 1u;
    __syncthreads();
    uint32_t *out = ml.nib + (((size_t)part * n_spectra + t) * (size_t)nranges + r) * NIB_WORDS;
    for (int i = tid; i < words; i += blockDim.x) out[i] = lw[i];
}

// adds the parts of a spectrum (8 bins per thread and step) to the ring spectrum; a spectrum flagged in nib_over
// is left to the exact recount
__global__ __launch_bounds__(256) void k_nibble_merge(MinimizerList ml, uint32_t *__restrict__ hists, MinimizerParams P,
                                                      uint32_t n_spectra, uint32_t n_parts, int nranges) {
    const int t = blockIdx.y;
    if (ml.nib_over[t]) return;
    const uint32_t slot = P.interval ? (uint32_t)(((uint64_t)t + P.ring_base) % P.ring_n) : P.ring_base;
    uint32_t *hist = hists + (size_t)slot * (size_t)P.num_bins;
    const int total_words = nranges * NIB_WORDS;
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < total_words; w += gridDim.x * blockDim.x) {
        const int r = w / NIB_WORDS, wi = w - r * NIB_WORDS;
        const int32_t b0 = r * NIB_BINS + wi * 8;
        if (b0 >= P.num_bins) continue;
        uint32_t even = 0, odd = 0, c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        uint32_t pending = 0;
        const size_t pstride = (size_t)n_spectra * (size_t)nranges * NIB_WORDS;
        const uint32_t *src = ml.nib + ((size_t)t * (size_t)nranges + r) * NIB_WORDS + wi;
        for (uint32_t p = 0; p < n_parts; p++) {
            uint32_t v = src[(size_t)p * pstride];
            if ((p & 3u) == 0 && p + 3 < n_parts) {                           // four independent loads in flight
                const uint32_t v1 = src[(size_t)(p + 1) * pstride], v2 = src[(size_t)(p + 2) * pstride], v3 = src[(size_t)(p + 3) * pstride];
                even += (v & 0x0F0F0F0Fu) + (v1 & 0x0F0F0F0Fu) + (v2 & 0x0F0F0F0Fu);
                odd += ((v >> 4) & 0x0F0F0F0Fu) + ((v1 >> 4) & 0x0F0F0F0Fu) + ((v2 >> 4) & 0x0F0F0F0Fu);
                v = v3; p += 3; pending += 3;
            }
            even += v & 0x0F0F0F0Fu; odd += (v >> 4) & 0x0F0F0F0Fu;          // byte lanes: at most 17 parts before widening
            if (++pending >= 14u || p + 1 == n_parts) {
#pragma unroll
                for (int q = 0; q < 4; q++) { c[2 * q] += (even >> (8 * q)) & 0xFFu; c[2 * q + 1] += (odd >> (8 * q)) & 0xFFu; }
                even = odd = 0; pending = 0;
            }
        }
#pragma unroll
        for (int q = 0; q < 8; q++) if (c[q] && b0 + q < P.num_bins) hist[b0 + q] += c[q];
    }
}

__global__ __launch_bounds__(256) void k_merge_hist(const uint32_t *__restrict__ partial, uint32_t *__restrict__ hists,
                                                    MinimizerParams P, uint32_t n_spectra, uint32_t n_parts,
                                                    const uint32_t *__restrict__ only_if) {
    const int t = blockIdx.y;
    if (only_if && !only_if[t]) return;
    const uint32_t slot = P.interval ? (uint32_t)(((uint64_t)t + P.ring_base) % P.ring_n) : P.ring_base;
    uint32_t *hist = hists + (size_t)slot * (size_t)P.num_bins;
    for (int32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < P.num_bins; b += gridDim.x * blockDim.x) {
        uint32_t v = 0;
        for (uint32_t p = 0; p < n_parts; p++) v += partial[((size_t)p * n_spectra + t) * (size_t)P.num_bins + b];
        if (v) hist[b] += v;
    }
}

// ------------------------------------------------------------------------------------------
// Long sequences (long reads, FASTA contigs: anything beyond the generic kernel's 1024 k-mer positions).
// A launch covers a GROUP of sequences (blockIdx.y = sequence of the group, blockIdx.x strides over its
// positions), each with its own slice of the scratch arrays and of the set table (LongSeqDesc), so
// 10-kb reads are not launch-bound.  k_long_hash: hashed canonical k-mer per position with the literal
// recurrence (N-safe); k_long_emit: windowed minimum per position; per-sequence set = open-addressing
// table in HBM (64-bit compare-and-swap), tried only where the window minimum differs from the one the
// previous position emitted (same set, ~5x fewer atomics); new values are jump-hashed and counted.
// ------------------------------------------------------------------------------------------
constexpr int LONG_PPT = 8;        // consecutive positions per thread in the long-sequence kernels
__global__ __launch_bounds__(256) void k_long_hash(const uint8_t *__restrict__ bases, const LongSeqDesc *__restrict__ desc,
                                                   MinimizerParams P, uint64_t *__restrict__ Xs_all,
                                                   uint8_t *__restrict__ valid_all) {
    __shared__ uint8_t lut[256];
    for (int t = threadIdx.x; t < 256; t += blockDim.x) lut[t] = nt4_of((unsigned)t);
    __syncthreads();
    const LongSeqDesc d = desc[blockIdx.y];
    const uint8_t *seq = bases + d.seq_off;
    uint64_t *Xs = Xs_all + d.xs_off;
    uint8_t *valid = valid_all + d.xs_off;
    const int64_t k = (int64_t)P.k, w = (int64_t)P.w;
    const uint64_t mask = (1ull << (2 * k)) - 1, shift = (uint64_t)(2 * (k - 1));
    const uint64_t npos = d.L - (uint64_t)k + 1;
    // a thread rolls through LONG_PPT consecutive positions: k + LONG_PPT bases instead of LONG_PPT * (k + 1)
    for (uint64_t j0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * LONG_PPT; j0 < npos;
         j0 += (uint64_t)gridDim.x * blockDim.x * LONG_PPT) {
        const uint64_t jend = j0 + LONG_PPT < npos ? j0 + LONG_PPT : npos;
        uint64_t f = 0, r = 0;
        // bases before (first position) - 1 cannot survive in f (masked) or r (shifted out): see k_minimizer_bin
        for (int64_t p = j0 > 0 ? (int64_t)j0 - 1 : 0; p < (int64_t)jend + k - 1; p++) {
            const uint64_t c = lut[seq[p]];
            f = (f << 2 | c) & mask;
            r = (r >> 2) | ((3ull ^ c) << shift);
            const int64_t j = p - (k - 1);
            if (j < (int64_t)j0) continue;
            uint64_t X = X_NONE; uint8_t ok = 0;
            if (f != r) {
                const uint64_t canon = f > r ? r : f;
                int64_t span = p - w + 2;
                if (span >= k) span = k;
                X = hash64(canon, mask) << 8 | (uint64_t)(int64_t)(int32_t)span;
                ok = 1;
            }
            Xs[j] = X; valid[j] = ok;
        }
    }
}

__global__ __launch_bounds__(256) void k_long_emit(const LongSeqDesc *__restrict__ desc, const uint64_t *__restrict__ Xs_all,
                                                   const uint8_t *__restrict__ valid_all, MinimizerParams P,
                                                   uint64_t *__restrict__ table_all, uint32_t *__restrict__ hists,
                                                   unsigned long long *__restrict__ min_slots) {
    __shared__ unsigned red[4];
    const LongSeqDesc d = desc[blockIdx.y];
    const uint64_t *Xs = Xs_all + d.xs_off;
    const uint8_t *valid = valid_all + d.xs_off;
    uint64_t *table = table_all + d.tab_off;
    const uint64_t table_mask = d.tab_mask;
    uint32_t *hist = hists + (size_t)d.hslot * (size_t)P.num_bins;
    const int64_t k = (int64_t)P.k, w = (int64_t)P.w;
    const uint64_t wwin = (uint64_t)(w > 0 ? w : 1);
    const uint64_t npos = d.L - (uint64_t)k + 1;
    unsigned fresh = 0;
    for (uint64_t j0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * LONG_PPT; j0 < npos;
         j0 += (uint64_t)gridDim.x * blockDim.x * LONG_PPT) {
        const uint64_t jend = j0 + LONG_PPT < npos ? j0 + LONG_PPT : npos;
        // sliding window minimum: m = min Xs[max(j-w+1,0) .. j]; a full rescan only when the value that
        // leaves the window is the current minimum (probability ~1/w per step)
        uint64_t m = X_NONE, mprev = X_NONE;
        bool prev_emit = false;
        if (j0 > 0) {                                           // window of position j0 - 1
            const uint64_t q = j0 - 1, lo = q >= wwin - 1 ? q - (wwin - 1) : 0;
            for (uint64_t p = lo; p <= q; p++) { const uint64_t x = Xs[p]; m = x < m ? x : m; }
            prev_emit = valid[q] && (int64_t)q + k - 1 >= w - 1;
            mprev = m;
        }
        for (uint64_t j = j0; j < jend; j++) {
            const uint64_t x = Xs[j];
            if (j >= wwin && Xs[j - wwin] == m) {               // the minimum leaves: rescan
                m = x;
                for (uint64_t p = j - wwin + 1; p < j; p++) { const uint64_t y = Xs[p]; m = y < m ? y : m; }
            } else {
                m = x < m ? x : m;
            }
            const bool emit = valid[j] && (int64_t)j + k - 1 >= w - 1;
            // the reference inserts the window minimum into the read's set at every emitting position; the
            // previous position already inserted the same value if it emitted with the same minimum
            if (emit && !(prev_emit && mprev == m)) {
                // per-read set: only the thread whose compare-and-swap claims the slot counts the value
                uint64_t slot = (m ^ (m >> 29)) * 0x9E3779B97F4A7C15ull >> 20 & table_mask;
                for (;;) {
                    const unsigned long long old = atomicCAS((unsigned long long *)&table[slot], (unsigned long long)TAB_EMPTY,
                                                             (unsigned long long)m);
                    if (old == TAB_EMPTY) { atomicAdd(&hist[jump_hash(m, P.num_bins)], 1u); fresh++; break; }
                    if (old == m) break;
                    slot = (slot + 1) & table_mask;
                }
            }
            prev_emit = emit; mprev = m;
        }
    }
    for (int off = 32; off; off >>= 1) fresh += __shfl_xor(fresh, off);
    if (lane_id() == 0) red[threadIdx.x >> 6] = fresh;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = red[0] + red[1] + red[2] + red[3];
        if (t) atomicAdd(&min_slots[(blockIdx.x + 131u * blockIdx.y) & (MIN_SLOTS - 1)], (unsigned long long)t);
    }
}

__global__ void k_fill_u64(uint64_t *p, uint64_t n, uint64_t v) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = v;
}

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

// K2: number of used bins (bitvector PopCount in the reference).  grid = (blocks, count)
__global__ __launch_bounds__(256) void k_count_used(const uint32_t *__restrict__ hists, DevState *st,
                                                    FlushBatch fb) {
    __shared__ unsigned red[4];
    const int t = blockIdx.y;
    const uint32_t slot = ring_slot(fb, t);
    if (blockIdx.x == 0 && t == 0 && threadIdx.x < RING_MAX) st->used[fb.parity ^ 1][threadIdx.x] = 0;  // arm the next flush
    const uint32_t *hist = hists + (size_t)slot * (size_t)fb.num_bins;
    unsigned cnt = 0;
    for (int32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < fb.num_bins; b += gridDim.x * blockDim.x)
        cnt += hist[b] != 0;
    for (int off = 32; off; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (lane_id() == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        cnt = red[0] + red[1] + red[2] + red[3];
        if (cnt) atomicAdd(&st->used[fb.parity][slot], cnt);   // one per block; the grid is small
    }
}



// ------------------------------------------------------------------------------------------
// K3 (no decay), bin-order form.  The chain-order kernels above gather 4-byte values along chains
// whose bins are ~2000 apart: rocprofv3 showed 13x more HBM traffic than the algorithmic bytes.
// Here every array is read in BIN order (coalesced) and the 7 x 2000 running counters live in LDS:
//   k_cms_segsum : per (spectrum, row, bin segment) sums per counter            (LDS atomics)
//   k_cms_base   : counter value in front of every (spectrum, segment)          (tiny prefix kernel)
//   k_cms_freq   : one workgroup per (segment, spectrum): waves 0..6 replay their row in bin order —
//                  est = ctr[pos] + (own + earlier same-counter bins of the 64-bin chunk, followed
//                  through a static "previous lane with the same counter" table) — wave 7 takes the
//                  minimum over the rows, writes f / 1/f and wipes the spectrum.  No est arrays at all.
// ------------------------------------------------------------------------------------------
constexpr int CMS_SEGS = 16;          // bin segments per spectrum
constexpr int CMS_GROUP = 4;          // 64-bin chunks staged per barrier

__global__ __launch_bounds__(512) void k_cms_segsum(const uint32_t *__restrict__ hists,
                                                    const uint16_t *__restrict__ pos16,
                                                    uint32_t *__restrict__ segsum, int depth, int width,
                                                    int seg_chunks, const DevState *st, FlushBatch fb) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *lctr = (uint32_t *)smem;                           // [depth][width]
    const int seg = blockIdx.x, t = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, d = tid >> 6;  // wave d = row d (depth waves + 1 idle)
    const uint32_t gomask = batch_gomask(st, fb);
    if (!((gomask >> t) & 1u)) return;
    for (int i = tid; i < depth * width; i += blockDim.x) lctr[i] = 0;
    __syncthreads();
    const size_t B = (size_t)fb.num_bins;
    const uint32_t *hist = hists + (size_t)ring_slot(fb, t) * B;
    if (d < depth) {
        const uint16_t *pd = pos16 + (size_t)d * B;
        const int64_t b0 = (int64_t)seg * seg_chunks * 64;
        for (int c0 = 0; c0 < seg_chunks; c0 += 8) {              // 8 chunks of loads in flight
            uint32_t h[8]; uint32_t p[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int64_t b = b0 + (int64_t)(c0 + u) * 64 + lane;
                const bool ok = (c0 + u < seg_chunks) && b < (int64_t)B;
                h[u] = ok ? hist[b] : 0u; p[u] = ok ? pd[b] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) if (h[u]) atomicAdd(&lctr[d * width + p[u]], h[u]);
        }
    }
    __syncthreads();
    uint32_t *out = segsum + (((size_t)t * depth) * CMS_SEGS + 0) * width;
    for (int i = tid; i < depth * width; i += blockDim.x) {
        const int dd = i / width, p = i - dd * width;
        out[((size_t)dd * CMS_SEGS + seg) * width + p] = lctr[i];
    }
}

// base[t][d][seg][p] = counter (d,p) in front of segment seg of spectrum t; advances the persistent counters
__global__ __launch_bounds__(256) void k_cms_base(const uint32_t *__restrict__ segsum,
                                                  unsigned long long *__restrict__ ctr,
                                                  unsigned long long *__restrict__ base, int depth, int width,
                                                  const DevState *st, FlushBatch fb) {
    const uint32_t gomask = batch_gomask(st, fb);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= depth * width) return;
    const int d = i / width, p = i - d * width;
    unsigned long long run = ctr[i];
    for (int t = 0; t < (int)fb.count; t++) {
        if (!((gomask >> t) & 1u)) continue;
        uint32_t sv[CMS_SEGS];
        const size_t at0 = (((size_t)t * depth + d) * CMS_SEGS) * width + p;
#pragma unroll
        for (int seg = 0; seg < CMS_SEGS; seg++) sv[seg] = segsum[at0 + (size_t)seg * width];   // independent loads
#pragma unroll
        for (int seg = 0; seg < CMS_SEGS; seg++) { base[at0 + (size_t)seg * width] = run; run += sv[seg]; }
    }
    ctr[i] = run;
}

__global__ __launch_bounds__(512) void k_cms_freq(uint32_t *__restrict__ hists, const uint16_t *__restrict__ pos16,
                                                  const uint8_t *__restrict__ meta8,
                                                  const unsigned long long *__restrict__ base,
                                                  double *__restrict__ f64, float *__restrict__ rcp32,
                                                  int depth, int width, int seg_chunks, size_t row_stride,
                                                  DevState *st, FlushBatch fb) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned long long *lctr = (unsigned long long *)smem;                       // [depth][width]
    unsigned long long *stage = lctr + (size_t)depth * width;                    // [2][depth][CMS_GROUP*64]
    const int seg = blockIdx.x, t = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, d = tid >> 6;                  // waves 0..depth-1: rows; wave depth: combiner
    const uint32_t gomask = batch_gomask(st, fb);
    const bool go = (gomask >> t) & 1u;
    const uint32_t slot = ring_slot(fb, t);
    if (seg == 0 && tid == 0) {
        const unsigned used = st->used[fb.parity][slot];
        if (used != 0 && !go) set_error(st, -5);                                 // "not used yet" (kmerspectrum.go:94-96)
        if (go) atomicAdd(&st->n_elements, (unsigned long long)used);
    }
    if (!go) return;
    const size_t B = (size_t)fb.num_bins;
    if (st->skip_exact[fb.parity]) {                                             // k_flush_decide: only Wipe is left to do
        uint32_t *hw = hists + (size_t)slot * B;
        const int64_t w0 = (int64_t)seg * seg_chunks * 64, w1 = w0 + (int64_t)seg_chunks * 64;
        for (int64_t b = w0 + tid; b < w1 && b < (int64_t)B; b += blockDim.x) hw[b] = 0;
        return;
    }
    {
        const unsigned long long *bt = base + (((size_t)t * depth) * CMS_SEGS) * width;
        for (int i = tid; i < depth * width; i += blockDim.x) {
            const int dd = i / width, p = i - dd * width;
            lctr[i] = bt[((size_t)dd * CMS_SEGS + seg) * width + p];
        }
    }
    __syncthreads();
    uint32_t *hist = hists + (size_t)slot * B;
    double *ft = f64 + (size_t)t * B;
    float *rt = rcp32 + (size_t)t * row_stride;
    const int64_t b0 = (int64_t)seg * seg_chunks * 64;
    const int ngroups = (seg_chunks + CMS_GROUP - 1) / CMS_GROUP;
    constexpr int GB = CMS_GROUP * 64;
    // software pipeline: the loads of group g+1 are issued before group g is processed
    const uint16_t *pd = pos16 + (size_t)(d < depth ? d : 0) * B;
    const uint8_t *md = meta8 + (size_t)(d < depth ? d : 0) * B;
    uint32_t nh[CMS_GROUP], np_[CMS_GROUP], nm[CMS_GROUP];
    auto load_group = [&](int g) {
#pragma unroll
        for (int c = 0; c < CMS_GROUP; c++) {
            const int ch = g * CMS_GROUP + c;
            const int64_t b = b0 + (int64_t)ch * 64 + lane;
            const bool ok = g < ngroups && ch < seg_chunks && b < (int64_t)B;
            nh[c] = ok ? hist[b] : 0u;
            if (d < depth) { np_[c] = ok ? pd[b] : 0u; nm[c] = ok ? md[b] : (64u | 0x80u); }
        }
    };
    load_group(0);
    uint32_t ph[CMS_GROUP];                                      // combiner: spectrum values of the group it finishes next
#pragma unroll
    for (int c = 0; c < CMS_GROUP; c++) ph[c] = 0;
    for (int g = 0; g <= ngroups; g++) {
        // rows: stage group g        combiner: finish group g-1
        uint32_t hh[CMS_GROUP], pp[CMS_GROUP], mm[CMS_GROUP];
#pragma unroll
        for (int c = 0; c < CMS_GROUP; c++) { hh[c] = nh[c]; pp[c] = np_[c]; mm[c] = nm[c]; }
        load_group(g + 1);
        if (d < depth && g < ngroups) {
            unsigned long long *my = stage + ((size_t)(g & 1) * depth + d) * GB;
            unsigned long long *rc = lctr + (size_t)d * width;
#pragma unroll
            for (int c = 0; c < CMS_GROUP; c++) {
                const int ch = g * CMS_GROUP + c;
                if (ch >= seg_chunks) break;
                const int64_t b = b0 + (int64_t)ch * 64 + lane;
                const uint32_t h = hh[c], p = pp[c], m = mm[c];
                // own count + the counts of the earlier lanes of this chunk that share the counter
                uint32_t acc = h, cur = m & 0x7fu;
                while (__any((int)(cur < 64u))) {
                    const uint32_t oh = (uint32_t)__shfl((int)h, (int)(cur & 63u));
                    const uint32_t oc = (uint32_t)__shfl((int)m, (int)(cur & 63u)) & 0x7fu;
                    if (cur < 64u) { acc += oh; cur = oc; }
                }
                const unsigned long long est = rc[p] + acc;          // every lane reads before any lane writes
                my[c * 64 + lane] = est;
                if ((m & 0x80u) && b < (int64_t)B) rc[p] = est;      // last lane of the chunk for this counter
            }
        }
        if (d == depth && g > 0) {
            const unsigned long long *src = stage + ((size_t)((g - 1) & 1) * depth) * GB;
#pragma unroll
            for (int c = 0; c < CMS_GROUP; c++) {
                const int ch = (g - 1) * CMS_GROUP + c;
                if (ch >= seg_chunks) break;
                const int64_t b = b0 + (int64_t)ch * 64 + lane;
                if (b < (int64_t)B) {
                    if (ph[c]) {
                        unsigned long long mn = ~0ull;
                        for (int dd = 0; dd < depth; dd++) { const unsigned long long e = src[(size_t)dd * GB + c * 64 + lane]; mn = e < mn ? e : mn; }
                        const double f = (double)mn;
                        ft[b] = f; rt[b] = (float)(1.0 / f);
                        hist[b] = 0;                                 // Wipe (kmerspectrum.go:58-64)
                    } else { ft[b] = 0.0; rt[b] = __builtin_nanf(""); }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CMS_GROUP; c++) ph[c] = hh[c];          // group g is finished by the combiner at g+1
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// K3 with uniform scaling (0 < decay < 1), bin-order form.  Counter (d,p) right after stream element j
// is C(j) = w*C(j-1) + (v_j if element j hits it).  Over a bin segment holding elements [e0,e1):
//     C(e1-1) = w^(e1-e0) * C(e0-1) + sum{ v_j * w^(e1-1-j) : hits }
//   k_cmsd_segsum : the sum (one w^x per element, LDS fp64 atomics) and the factor per segment
//   k_cmsd_base   : C in front of every (spectrum, segment), advancing the persistent fp64 counters
//   k_cmsd_freq   : replay in bin order with lazily decayed LDS counters {value, time}; zero bins are
//                   transparent; same-counter lanes of a 64-bin chunk resolve in lane order
// (fp64 sums are re-associated w.r.t. the reference's step-by-step scaling: ~1e-13 relative)
// ------------------------------------------------------------------------------------------
// w^x for an element gap 0 <= x < 2^21 (a spectrum has < 2^20 elements): three 128-entry tables in LDS
// (w^a, w^(128 b), w^(16384 c)) and two multiplications instead of an fp64 exp() per counter update — 3 ulp.
constexpr int POWW_N = 384;
__device__ __forceinline__ void poww_build(double *tab, double lnw, int tid, int nthreads) {
    for (int i = tid; i < POWW_N; i += nthreads) {
        const int lvl = i >> 7, a = i & 127;
        tab[i] = exp((double)a * (lvl == 0 ? 1.0 : lvl == 1 ? 128.0 : 16384.0) * lnw);
    }
}
__device__ __forceinline__ double poww(const double *tab, long long x, double lnw) {
    if ((unsigned long long)x >= (1ull << 21)) return exp((double)x * lnw);
    const uint32_t u = (uint32_t)x;
    return tab[u & 127u] * tab[128u + ((u >> 7) & 127u)] * tab[256u + (u >> 14)];
}
constexpr int CMSD_GROUP = 2;         // chunks staged per barrier (LDS: 112 KB values + 28 KB times + 14 KB stage)

__global__ __launch_bounds__(512) void k_cmsd_segsum(const uint32_t *__restrict__ hists, const uint16_t *__restrict__ pos16,
                                                     const uint32_t *__restrict__ eidx, const uint32_t *__restrict__ etot,
                                                     double *__restrict__ segadd, double *__restrict__ segfac,
                                                     uint32_t *__restrict__ sege0, int depth, int width, int seg_chunks,
                                                     double omega, const DevState *st, FlushBatch fb) {
    extern __shared__ __align__(16) unsigned char smem[];
    double *ladd = (double *)smem;                                // [depth][width]
    const int seg = blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
    const uint32_t gomask = batch_gomask(st, fb);
    if (!((gomask >> t) & 1u)) return;
    for (int i = tid; i < depth * width; i += blockDim.x) ladd[i] = 0.0;
    __syncthreads();
    const size_t B = (size_t)fb.num_bins;
    const uint32_t *hist = hists + (size_t)ring_slot(fb, t) * B;
    const uint32_t *ei = eidx + (size_t)t * B;
    const int64_t b0 = (int64_t)seg * seg_chunks * 64, b1 = b0 + (int64_t)seg_chunks * 64;
    const uint32_t e0 = b0 < (int64_t)B ? ei[b0] : etot[t];
    const uint32_t e1 = b1 < (int64_t)B ? ei[b1] : etot[t];
    const double lnw = log(omega);                                // w^x = exp(x ln w): |x ln w| * 2^-53 relative, far below the tolerance
    if (tid == 0) { segfac[(size_t)t * CMS_SEGS + seg] = exp((double)(e1 - e0) * lnw); sege0[(size_t)t * CMS_SEGS + seg] = e0; }
    for (int64_t b = b0 + tid; b < b1 && b < (int64_t)B; b += blockDim.x) {
        const uint32_t h = hist[b];
        if (h) {
            const double wgt = (double)h * exp((double)(e1 - 1u - ei[b]) * lnw);
            for (int d = 0; d < depth; d++) atomicAdd(&ladd[d * width + pos16[(size_t)d * B + b]], wgt);
        }
    }
    __syncthreads();
    for (int i = tid; i < depth * width; i += blockDim.x) {
        const int dd = i / width, p = i - dd * width;
        segadd[((((size_t)t * depth) + dd) * CMS_SEGS + seg) * width + p] = ladd[i];
    }
}

__global__ __launch_bounds__(256) void k_cmsd_base(const double *__restrict__ segadd, const double *__restrict__ segfac,
                                                   double *__restrict__ ctrd, double *__restrict__ cstart, int depth,
                                                   int width, const DevState *st, FlushBatch fb) {
    const uint32_t gomask = batch_gomask(st, fb);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= depth * width) return;
    const int d = i / width, p = i - d * width;
    double C = ctrd[i];
    for (int t = 0; t < (int)fb.count; t++) {
        if (!((gomask >> t) & 1u)) continue;
        double sv[CMS_SEGS], fv[CMS_SEGS];
        const size_t at0 = (((size_t)t * depth + d) * CMS_SEGS) * width + p;
#pragma unroll
        for (int seg = 0; seg < CMS_SEGS; seg++) { sv[seg] = segadd[at0 + (size_t)seg * width]; fv[seg] = segfac[(size_t)t * CMS_SEGS + seg]; }
#pragma unroll
        for (int seg = 0; seg < CMS_SEGS; seg++) { cstart[at0 + (size_t)seg * width] = C; C = C * fv[seg] + sv[seg]; }
    }
    ctrd[i] = C;
}

__global__ __launch_bounds__(512) void k_cmsd_freq(uint32_t *__restrict__ hists, const uint16_t *__restrict__ pos16,
                                                   const uint8_t *__restrict__ meta8, const uint32_t *__restrict__ eidx,
                                                   const uint32_t *__restrict__ sege0, const double *__restrict__ cstart,
                                                   double *__restrict__ f64, float *__restrict__ rcp32, int depth,
                                                   int width, int seg_chunks, size_t row_stride, double omega,
                                                   DevState *st, FlushBatch fb) {
    extern __shared__ __align__(16) unsigned char smem[];
    double *lval = (double *)smem;                                               // [depth][width] counter value ...
    double *stage = lval + (size_t)depth * width;                                // [2][depth][CMSD_GROUP*64]
    uint16_t *ltime = (uint16_t *)(stage + (size_t)2 * depth * CMSD_GROUP * 64); // ... as of element e0 - 1 + ltime
    const int seg = blockIdx.x, t = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, d = tid >> 6;
    const uint32_t gomask = batch_gomask(st, fb);
    const bool go = (gomask >> t) & 1u;
    const uint32_t slot = ring_slot(fb, t);
    if (seg == 0 && tid == 0) {
        const unsigned used = st->used[fb.parity][slot];
        if (used != 0 && !go) set_error(st, -5);
        if (go) atomicAdd(&st->n_elements, (unsigned long long)used);
    }
    if (!go) return;
    const size_t B = (size_t)fb.num_bins;
    {
        const double *bt = cstart + (((size_t)t * depth) * CMS_SEGS) * width;
        for (int i = tid; i < depth * width; i += blockDim.x) {
            const int dd = i / width, p = i - dd * width;
            lval[i] = bt[((size_t)dd * CMS_SEGS + seg) * width + p];
            ltime[i] = 0;
        }
    }
    __syncthreads();
    uint32_t *hist = hists + (size_t)slot * B;
    const uint32_t *ei = eidx + (size_t)t * B;
    double *ft = f64 + (size_t)t * B;
    float *rt = rcp32 + (size_t)t * row_stride;
    const int64_t b0 = (int64_t)seg * seg_chunks * 64;
    const long long tref = (long long)sege0[(size_t)t * CMS_SEGS + seg] - 1;     // time of ltime == 0
    const int ngroups = (seg_chunks + CMSD_GROUP - 1) / CMSD_GROUP;
    constexpr int GB = CMSD_GROUP * 64;
    const double lnw = log(omega);
    __shared__ double pw[64];                                     // w^x for the gaps inside one 64-bin chunk
    __shared__ double pwt[POWW_N];                                // ... and for any gap (poww)
    if (tid < 64) pw[tid] = exp((double)tid * lnw);
    poww_build(pwt, lnw, tid, (int)blockDim.x);
    __syncthreads();
    const uint16_t *pd = pos16 + (size_t)(d < depth ? d : 0) * B;
    const uint8_t *md = meta8 + (size_t)(d < depth ? d : 0) * B;
    // the four per-bin inputs of group g+1 are requested before group g is computed: with one workgroup per CU
    // (LDS) and a barrier per group, their latency was the kernel's time
    uint32_t nh[CMSD_GROUP], np_[CMSD_GROUP], nm[CMSD_GROUP], nj[CMSD_GROUP];
    auto fetch = [&](int g) {
#pragma unroll
        for (int c = 0; c < CMSD_GROUP; c++) {
            const int ch = g * CMSD_GROUP + c;
            const int64_t b = b0 + (int64_t)ch * 64 + lane;
            nh[c] = 0; np_[c] = 0; nm[c] = 64u | 0x80u; nj[c] = 0;
            if (d <= depth && ch < seg_chunks && b < (int64_t)B) {
                nh[c] = hist[b];                                   // (the combiner wave, d == depth, needs only this one)
                if (d < depth) { np_[c] = pd[b]; nm[c] = md[b]; nj[c] = ei[b]; }
            }
        }
    };
    fetch(0);
    uint32_t ch_[CMSD_GROUP] = {}, cp[CMSD_GROUP], cm[CMSD_GROUP], cj[CMSD_GROUP], ph[CMSD_GROUP];
    for (int g = 0; g <= ngroups; g++) {
#pragma unroll
        for (int c = 0; c < CMSD_GROUP; c++) { ph[c] = ch_[c]; ch_[c] = nh[c]; cp[c] = np_[c]; cm[c] = nm[c]; cj[c] = nj[c]; }
        if (g + 1 < ngroups) fetch(g + 1);
        if (d < depth && g < ngroups) {
            double *my = stage + ((size_t)(g & 1) * depth + d) * GB;
            double *rv = lval + (size_t)d * width;
            uint16_t *rtm = ltime + (size_t)d * width;
#pragma unroll
            for (int c = 0; c < CMSD_GROUP; c++) {
                const int ch = g * CMSD_GROUP + c;
                if (ch >= seg_chunks) break;
                const int64_t b = b0 + (int64_t)ch * 64 + lane;
                const uint32_t h = ch_[c], p = cp[c], m = cm[c]; const long long j = (long long)cj[c];
                // resolve the lanes in same-counter order: a lane is computed once its predecessor is
                const uint32_t prev = m & 0x7fu;
                bool ready = false; double C = 0.0; long long tj = 0;
                if (prev >= 64u) {                                  // first lane of the chunk on this counter: LDS state
                    const double C0 = rv[p]; const long long t0 = tref + (long long)rtm[p];
                    if (h) { C = C0 * poww(pwt, j - t0, lnw) + (double)h; tj = j; } else { C = C0; tj = t0; }
                    ready = true;
                }
                while (__any((int)!ready)) {
                    const double pc = __shfl(C, (int)(prev & 63u));
                    const long long pt = __shfl(tj, (int)(prev & 63u));
                    const int pr = __shfl((int)ready, (int)(prev & 63u));
                    if (!ready && pr) {
                        if (h) { C = pc * pw[(int)(j - pt) & 63] + (double)h; tj = j; } else { C = pc; tj = pt; }   // gap < 64 inside a chunk
                        ready = true;
                    }
                }
                my[c * 64 + lane] = C;
                if ((m & 0x80u) && b < (int64_t)B) { rv[p] = C; rtm[p] = (uint16_t)(tj - tref); }
            }
        }
        if (d == depth && g > 0) {
            const double *src = stage + ((size_t)((g - 1) & 1) * depth) * GB;
            for (int c = 0; c < CMSD_GROUP; c++) {
                const int ch = (g - 1) * CMSD_GROUP + c;
                if (ch >= seg_chunks) break;
                const int64_t b = b0 + (int64_t)ch * 64 + lane;
                if (b < (int64_t)B) {
                    if (ph[c]) {                                   // hist[b], fetched two groups ago
                        double mn = INFINITY;
                        for (int dd = 0; dd < depth; dd++) { const double e = src[(size_t)dd * GB + c * 64 + lane]; mn = e < mn ? e : mn; }
                        ft[b] = mn; rt[b] = (float)(1.0 / mn);
                        hist[b] = 0;
                    } else { ft[b] = 0.0; rt[b] = __builtin_nanf(""); }
                }
            }
        }
        __syncthreads();
    }
}

// Wave-wide minima of 8 independent (non-NaN) values, transposed: 19 instructions instead of the 48
// of six DPP butterfly steps per value (and hipcc emits mov+mov_dpp+canonicalise+min per step for the
// equivalent builtins, 4x that again — hence one asm block).
// v_permlane32_swap / v_permlane16_swap (gfx950) exchange half-waves / 16-lane rows between two
// registers, so one swap + one v_min folds two rows' partial minima at once and halves the number of
// live registers: 8 -> 4 (halves) -> 2 (rows); the two survivors are folded over 8-lane halves with
// row_ror:8, merged into one register (lanes 8..15 of every row take the second) and finished with
// three DPP steps inside groups of 8 lanes.  Lane l returns the wave minimum of value l / 8.
__device__ __forceinline__ float wave_min8_by_row(float (&m)[8]) {
    // one block: the swaps need 2 wait states after a VALU write of either operand (s_nop), DPP sources too
    asm volatile("s_nop 1\n\t"
                 "v_permlane32_swap_b32 %0, %4\n\t"        // lanes 0..31: value j, lanes 32..63: value j+4
                 "v_permlane32_swap_b32 %1, %5\n\t"
                 "v_permlane32_swap_b32 %2, %6\n\t"
                 "v_permlane32_swap_b32 %3, %7\n\t"
                 "v_min_f32 %0, %0, %4\n\t"
                 "v_min_f32 %2, %2, %6\n\t"
                 "v_min_f32 %1, %1, %5\n\t"
                 "v_min_f32 %3, %3, %7\n\t"
                 "v_permlane16_swap_b32 %0, %2\n\t"        // 16-lane row q: value 2q
                 "s_nop 0\n\t"
                 "v_permlane16_swap_b32 %1, %3\n\t"        // 16-lane row q: value 2q+1
                 "v_min_f32 %0, %0, %2\n\t"
                 "v_min_f32 %1, %1, %3\n\t"
                 "s_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_min_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_mov_b32_dpp %0, %1 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xc\n\t"   // lanes 8..15 of each row
                 "s_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]));
    return m[0];
}

// Whole-batch bound (no concept drift).  Every count-min estimate of the batch is at least the smallest
// counter at batch start (an estimate is a minimum over counters that only grow), so A = K/f >= min_slot(K) / Cmin
// for every negative K of a slot's row (>= 0 otherwise).  If that cannot get below the current weight of ANY slot
// (same 1e-5 band as the scan), no AddElement of the batch can change the sketch: skip_exact is raised and
// k_cms_freq only wipes the spectra, k_rcp_extrema / k_cws_scan / k_cws_resolve / k_cws_apply return at once.
// The count-min counters are still advanced (k_cms_segsum + k_cms_base).  After the first intervals of a stream
// this is the normal case: counters are in the tens of thousands while the winning weights came from f ~ 10.
__global__ __launch_bounds__(1024) void k_flush_decide(const unsigned long long *__restrict__ ctr, int ncounters,
                                                       const float *__restrict__ kminslot,
                                                       const double *__restrict__ weights, int slots, int slot_begin,
                                                       DevState *st, FlushBatch fb, int enable) {
    __shared__ unsigned long long red[16];
    __shared__ int anypass;
    const int tid = threadIdx.x;
    if (tid == 0) anypass = 0;
    unsigned long long m = ~0ull;
    for (int i = tid; i < ncounters; i += blockDim.x) { const unsigned long long v = ctr[i]; m = v < m ? v : m; }
    for (int off = 32; off; off >>= 1) { const unsigned long long o = __shfl_xor(m, off); m = o < m ? o : m; }
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = red[0];
    for (int i = 1; i < 16; i++) m = red[i] < m ? red[i] : m;
    bool pass = false;
    if (!enable || m == 0) pass = true;                          // an untouched counter: estimates can be as small as 1
    else {
        const double rmax = 1.0 / (double)m;
        for (int s = tid; s < slots; s += blockDim.x) {
            const double km = (double)kminslot[s];
            const double w = weights[slot_begin + s];
            const double thr = w + 1e-5 * fabs(w) + 1e-37;
            const double bound = km < 0.0 ? km * rmax : 0.0;
            if (bound <= thr) pass = true;
        }
    }
    if (pass) atomicOr(&anypass, 1);
    __syncthreads();
    if (tid == 0) st->skip_exact[fb.parity] = anypass ? 0u : 1u;
}
__global__ __launch_bounds__(256) void k_slot_kmin(const float *__restrict__ kmin32, float *__restrict__ kminslot, int wtiles) {
    __shared__ float red[4];
    const int slot = blockIdx.x;
    float m = INFINITY;
    for (int i = threadIdx.x; i < wtiles; i += blockDim.x) m = fminf(m, kmin32[(size_t)slot * wtiles + i]);
    for (int off = 32; off; off >>= 1) m = fminf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) kminslot[slot] = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
}

// Static per (slot, 256-bin wave tile) minimum of K32, and per flush the extrema of the reciprocal
// vectors per (interval, wave tile): the inputs of k_cws_scan's bound test.
__global__ __launch_bounds__(256) void k_tile_kmin(const float *__restrict__ k32, float *__restrict__ kmin32,
                                                   int ntiles, size_t row_stride) {
    const int slot = blockIdx.y, tile = blockIdx.x, wid = threadIdx.x >> 6;
    const floatx4 v = *(const floatx4 *)(k32 + (size_t)slot * row_stride + (size_t)tile * SCAN_TILE + (size_t)threadIdx.x * 4);
    float m = fminf(fminf(v.x, v.y), fminf(v.z, v.w));
    for (int off = 32; off; off >>= 1) m = fminf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) kmin32[(size_t)slot * (size_t)(ntiles * 4) + (size_t)(tile * 4 + wid)] = m;
}
__global__ __launch_bounds__(256) void k_rcp_extrema(const float *__restrict__ rcp32, float *__restrict__ rext,
                                                     int ntiles, size_t row_stride, const DevState *st, FlushBatch fb) {
    if (st->skip_exact[fb.parity]) return;
    const int t = blockIdx.y, tile = blockIdx.x, wid = threadIdx.x >> 6;
    const floatx4 v = *(const floatx4 *)(rcp32 + (size_t)t * row_stride + (size_t)tile * SCAN_TILE + (size_t)threadIdx.x * 4);
    // NaN = bin not in the stream: fmaxf / fminf return the other operand
    float hi = fmaxf(fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)), -INFINITY);
    float lo = fminf(fminf(fminf(v.x, v.y), fminf(v.z, v.w)), INFINITY);
    for (int off = 32; off; off >>= 1) { hi = fmaxf(hi, __shfl_xor(hi, off)); lo = fminf(lo, __shfl_xor(lo, off)); }
    if ((threadIdx.x & 63) == 0) {
        float *o = rext + ((size_t)t * (size_t)(ntiles * 4) + (size_t)(tile * 4 + wid)) * 2;
        o[0] = hi; o[1] = lo;
    }
}

// ------------------------------------------------------------------------------------------
// K4a: the HBM-bound pass.  A[t][slot][bin] = K[slot][bin] * (1/f_t[bin]); minimum per
// (interval t, slot, 256-bin wave tile).  A workgroup streams SCAN_ROWS rows x SCAN_TILE bins of K
// exactly once (16 B per lane per row, SCAN_ROWS independent loads in flight) and re-uses the
// registers for every interval of the batch, so the table is read once per BATCH, not per interval.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cws_scan(const float *__restrict__ k32,
                                                  const float *__restrict__ rcp32,
                                                  float *__restrict__ tilemin, int slots, int ntiles,
                                                  size_t row_stride, const DevState *st, FlushBatch fb,
                                                  const float *__restrict__ kmin32, const float *__restrict__ rext,
                                                  const double *__restrict__ weights, int slot_begin,
                                                  unsigned long long *__restrict__ visited, double drift_dw) {
    // XCD-aware order (workgroup b lands on XCD b % 8): XCD x works through column tile 8*chunk + x for
    // ALL slot groups before moving on, so a column's reciprocal vectors (T x 4 KB) are fetched into
    // that XCD's L2 once and re-used by the other groups; the 8 XCDs stream 8 adjacent 4 KB pieces of
    // the same K rows at the same time (32 KB contiguous per row).
    if (st->skip_exact[fb.parity]) return;                       // k_flush_decide: nothing in this batch can matter
    const int ngrp = (slots + SCAN_ROWS - 1) / SCAN_ROWS;
    const int chunk = blockIdx.x / (8 * ngrp), rem = blockIdx.x % (8 * ngrp);
    const int grp = rem / 8, tile = chunk * 8 + (rem % 8);
    if (tile >= ntiles) return;
    const int tid = threadIdx.x, wid = tid >> 6;
    const size_t col = (size_t)tile * SCAN_TILE + (size_t)tid * 4;
    const int wtiles = ntiles * 4;                               // 256-bin wave tiles per row
    const uint32_t gomask = batch_gomask(st, fb);
    const int ngroups = (slots + SCAN_ROWS - 1) / SCAN_ROWS;
    const int lane = tid & 63;
    // ---- branch and bound (no concept drift): AddElement only ever replaces a slot's weight by a SMALLER
    // A, and A = K * (1/f) >= min(K) * max(1/f) over a tile (min(K) < 0; min(K) * min(1/f) otherwise).  A wave
    // tile whose bound cannot get below the slot's current weight for any row and interval of the batch is
    // not read at all — after the first intervals of a stream that is nearly every tile, because count-min
    // estimates only grow.  The weights at batch start are used (they only fall during the batch), with the
    // same 1e-5 relative band the fp64 resolve uses around fp32 values; skipped tiles report +inf.
    if (kmin32) {
        const int wt = tile * 4 + wid, row = lane & 7, slot = grp * SCAN_ROWS + row;
        bool pass = false;
        if (slot < slots) {
            const double km = (double)kmin32[(size_t)slot * wtiles + wt];
            double w = weights[slot_begin + slot];
            // concept drift (drift_dw = decayWeight > 0): the update test is A < w / decayWeight and w may move either
            // way — but a NEGATIVE weight can only be replaced by a smaller one (A < w/dw < w), so its threshold of
            // the whole batch is at most w_start / dw; a slot whose weight is not negative is simply never pruned
            bool never = false;
            if (drift_dw > 0.0) { if (w < 0.0) w = w / drift_dw; else never = true; }
            const double thr = w + 1e-5 * fabs(w) + 1e-37;
            if (never) pass = true;
            for (int t = lane >> 3; t < (int)fb.count; t += 8) {
                if (!((gomask >> t) & 1u)) continue;
                const float rmax = rext[((size_t)t * wtiles + wt) * 2], rmin = rext[((size_t)t * wtiles + wt) * 2 + 1];
                if (!(rmax > 0.f)) continue;                     // no element of this interval falls into the tile
                const double bound = km < 0.0 ? km * (double)rmax : km * (double)rmin;
                if (bound <= thr) pass = true;
            }
        }
        if (!__ballot(pass)) {
            for (int t = lane >> 3; t < (int)fb.count; t += 8)
                tilemin[(((size_t)t * ngroups + grp) * wtiles + (size_t)wt) * SCAN_ROWS + row] = INFINITY;
            return;
        }
        if (lane == 0) atomicAdd(&visited[blockIdx.x & (MIN_SLOTS - 1)], 1ull);
    }
    floatx4 kv[SCAN_ROWS];
#pragma unroll
    for (int r = 0; r < SCAN_ROWS; r++) {
        const int slot = grp * SCAN_ROWS + r;
        if (slot < slots)
            kv[r] = __builtin_nontemporal_load((const floatx4 *)(k32 + (size_t)slot * row_stride + col));
        else
            kv[r] = (floatx4)(0.f);
    }
    floatx4 rc_next = *(const floatx4 *)(rcp32 + col);
    for (int t = 0; t < (int)fb.count; t++) {
        const floatx4 rc = rc_next;
        if (t + 1 < (int)fb.count) rc_next = *(const floatx4 *)(rcp32 + (size_t)(t + 1) * row_stride + col);
        if (!((gomask >> t) & 1u)) continue;
        float m[SCAN_ROWS];
        // v_mul_f32 x4 + v_min3_f32 x2 per row.  (v_pk_mul_f32 halves the multiplies but runs this loop 2x
        // SLOWER on MI355X — measured 304 vs 150 us — so the products stay scalar.)  NaN (bin not in the
        // stream) loses every v_min; INFINITY keeps an all-NaN lane out of the reduction.
#pragma unroll
        for (int r = 0; r < SCAN_ROWS; r++)
            m[r] = fminf(fminf(fminf(fminf(kv[r].x * rc.x, kv[r].y * rc.y), kv[r].z * rc.z), kv[r].w * rc.w), INFINITY);
        static_assert(SCAN_ROWS == 8, "wave_min8_by_row reduces exactly 8 rows");
        const float mine = wave_min8_by_row(m);                    // lane l: minimum of row l / 8 over the wave
        // tilemin[t][slot group][wave tile][row]: 8 lanes write the 8 rows of a wave tile (32 contiguous bytes)
        if ((lane & 7) == 0)
            tilemin[(((size_t)t * ngroups + grp) * wtiles + (size_t)(tile * 4 + wid)) * SCAN_ROWS + (lane >> 3)] = mine;
    }
}

// ------------------------------------------------------------------------------------------
// K4b: per slot, interval by interval: re-evaluate in fp64 — with the literal getSample formula —
// every wave tile whose fp32 minimum is within a relative band of the slot's fp32 minimum, then
// apply AddElement's update rule.  The band (1e-5 rel + 1e-37 abs) is >30x the worst fp32 error
// of K4a, so the true fp64 argmin (and every exact tie, for earliest-wins) is always inside a
// re-evaluated tile.
// ------------------------------------------------------------------------------------------
constexpr int WTILE = SCAN_TILE / 4;
__global__ __launch_bounds__(256) void k_cws_resolve(const double *__restrict__ rcb,
                                                     const double *__restrict__ f64,
                                                     const float *__restrict__ tilemin,
                                                     double *__restrict__ candA, int32_t *__restrict__ candB,
                                                     int slots, int ntiles, const DevState *st,
                                                     FlushBatch fb) {
    extern __shared__ __align__(16) unsigned char smem[];
    float *tm = (float *)smem;                       // [wtiles]
    __shared__ float redf[4];
    __shared__ double redA[4];
    __shared__ int32_t redB[4];
    __shared__ int ncand;
    __shared__ int cand[64];
    if (st->skip_exact[fb.parity]) return;
    const int slot = blockIdx.x, t = blockIdx.y;     // local slot, interval of the batch
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wtiles = ntiles * 4;
    const int32_t num_bins = fb.num_bins;
    double bestA = INFINITY; int32_t bestB = 0x7fffffff;
    if (flush_go(st, fb, t)) {                        // block-uniform
        const double *row = rcb + (size_t)slot * (size_t)num_bins * 3;
        const int ngroups = (slots + SCAN_ROWS - 1) / SCAN_ROWS;
        const float *tmin_t = tilemin + (((size_t)t * ngroups + slot / SCAN_ROWS) * wtiles) * SCAN_ROWS + (slot % SCAN_ROWS);
        const double *ft = f64 + (size_t)t * (size_t)num_bins;
        if (tid == 0) ncand = 0;
        float g = INFINITY;
        for (int x = tid; x < wtiles; x += blockDim.x) {
            const float v = tmin_t[(size_t)x * SCAN_ROWS];
            tm[x] = v;
            g = fminf(g, v);
        }
        for (int off = 32; off; off >>= 1) g = fminf(g, __shfl_xor(g, off));
        if (lane == 0) redf[wid] = g;
        __syncthreads();
        g = fminf(fminf(redf[0], redf[1]), fminf(redf[2], redf[3]));
        if (g < INFINITY) {                           // some element reached this slot's row
            const float thr = g + 1e-5f * fabsf(g) + 1e-37f;
            // candidate wave tiles (normally one): every tile whose fp32 minimum is inside the band
            bool overflow = false;
            for (int x = tid; x < wtiles; x += blockDim.x)
                if (tm[x] <= thr) { const int at = atomicAdd(&ncand, 1); if (at < 64) cand[at] = x; else overflow = true; }
            __syncthreads();
            const int nc = ncand;                     // block-uniform
            if (nc <= 64) {
                for (int ci = 0; ci < nc; ci++) {
                    const int32_t bin = cand[ci] * WTILE + tid;      // WTILE == blockDim.x
                    if (bin < num_bins) {
                        const double f = ft[bin];
                        if (f != 0.0) {
                            const double r = row[(size_t)bin * 3 + 0];
                            const double c = row[(size_t)bin * 3 + 1];
                            const double b = row[(size_t)bin * 3 + 2];
                            const double Yka = exp(log(f) - b);
                            const double A = c / (Yka * exp(r));
                            if (A < bestA || (A == bestA && bin < bestB)) { bestA = A; bestB = bin; }
                        }
                    }
                }
            } else {
                // degenerate spectrum (many equal minima): evaluate every tile inside the band
                for (int x = 0; x < wtiles; x++) {
                    if (!(tm[x] <= thr)) continue;
                    const int32_t bin = x * WTILE + tid;
                    if (bin < num_bins) {
                        const double f = ft[bin];
                        if (f != 0.0) {
                            const double r = row[(size_t)bin * 3 + 0];
                            const double c = row[(size_t)bin * 3 + 1];
                            const double b = row[(size_t)bin * 3 + 2];
                            const double Yka = exp(log(f) - b);
                            const double A = c / (Yka * exp(r));
                            if (A < bestA || (A == bestA && bin < bestB)) { bestA = A; bestB = bin; }
                        }
                    }
                }
            }
            (void)overflow;
            for (int off = 32; off; off >>= 1) {
                const double oA = __shfl_xor(bestA, off);
                const int32_t oB = __shfl_xor(bestB, off);
                if (oA < bestA || (oA == bestA && oB < bestB)) { bestA = oA; bestB = oB; }
            }
            if (lane == 0) { redA[wid] = bestA; redB[wid] = bestB; }
            __syncthreads();
#pragma unroll
            for (int x = 0; x < 4; x++)
                if (redA[x] < bestA || (redA[x] == bestA && redB[x] < bestB)) { bestA = redA[x]; bestB = redB[x]; }
        }
    }
    if (tid == 0) { candA[(size_t)t * slots + slot] = bestA; candB[(size_t)t * slots + slot] = bestB; }
}

// AddElement's slot update (histosketch.go:150-153, no drift) in interval order: the element with
// the smallest A of interval t replaces the slot iff it is strictly below the running weight.
__global__ void k_cws_apply(const double *__restrict__ candA, const int32_t *__restrict__ candB,
                            unsigned long long *__restrict__ mins, double *__restrict__ weights,
                            int slots, int slot_begin, const DevState *st, FlushBatch fb) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= slots || st->skip_exact[fb.parity]) return;
    const int gs = slot_begin + slot;
    double w = weights[gs]; unsigned long long m = mins[gs];
    for (int t = 0; t < (int)fb.count; t++) {
        if (!flush_go(st, fb, t)) continue;
        const double A = candA[(size_t)t * slots + slot];
        const int32_t b = candB[(size_t)t * slots + slot];
        if (b != 0x7fffffff && A < w) { w = A; m = (unsigned long long)b; }
    }
    weights[gs] = w; mins[gs] = m;
}

// ==========================================================================================
// Concept drift (decay_ratio != 1): reference src/countmin/countmin.go:49-56,103-110,141-147 and
// src/histosketch/histosketch.go:79-81,139-153.
//
// Count-min with uniform scaling (0 < decay < 1): every Add() first multiplies ALL counters by
// w = exp(-decay).  With i = index of an element inside its flush (ascending non-zero bins), the
// counter (d,g) right after element i is  C0*w^(i+1) + sum{ v_j * w^(i-j) : j <= i, pos_d(b_j)=g },
// i.e. along a chain the first-order recurrence  S_m = w^(gap_m) * S_{m-1} + v_m  — an affine scan.
// (The reference multiplies step by step; the closed form differs by accumulated rounding of
// ~gap*2^-53 relative, far inside the 1e-5 tolerance the north star states for CWS values.)
// ==========================================================================================

// element index of every bin = number of non-zero bins in front of it.  grid = (blocks, count)
constexpr int EIDX_BLOCK = 2048;
__global__ __launch_bounds__(256) void k_elem_count(const uint32_t *__restrict__ hists,
                                                    uint32_t *__restrict__ blkcnt, int nblk, FlushBatch fb) {
    __shared__ unsigned red[4];
    const int t = blockIdx.y, blk = blockIdx.x;
    const uint32_t *hist = hists + (size_t)ring_slot(fb, t) * (size_t)fb.num_bins;
    unsigned cnt = 0;
    for (int x = 0; x < EIDX_BLOCK / 256; x++) {
        const int32_t b = blk * EIDX_BLOCK + x * 256 + threadIdx.x;
        if (b < fb.num_bins) cnt += hist[b] != 0;
    }
    for (int off = 32; off; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (lane_id() == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) blkcnt[(size_t)t * nblk + blk] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void k_elem_index(const uint32_t *__restrict__ hists,
                                                    const uint32_t *__restrict__ blkcnt,
                                                    uint32_t *__restrict__ eidx, uint32_t *__restrict__ etot,
                                                    int nblk, FlushBatch fb) {
    __shared__ unsigned red[4];
    __shared__ unsigned wsum[4];
    const int t = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const size_t B = (size_t)fb.num_bins;
    const uint32_t *hist = hists + (size_t)ring_slot(fb, t) * B;
    // offset of this block = sum of the counts of the blocks in front of it
    unsigned off = 0, all = 0;
    for (int x = tid; x < nblk; x += 256) { const unsigned c = blkcnt[(size_t)t * nblk + x]; all += c; if (x < blk) off += c; }
    for (int o = 32; o; o >>= 1) { off += __shfl_xor(off, o); all += __shfl_xor(all, o); }
    if (lane == 0) { red[wid] = off; wsum[wid] = all; }
    __syncthreads();
    off = red[0] + red[1] + red[2] + red[3];
    if (blk == 0 && tid == 0) etot[t] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    // 8 consecutive bins per thread
    const int32_t b0 = blk * EIDX_BLOCK + tid * 8;
    unsigned nz[8]; unsigned mine = 0;
#pragma unroll
    for (int x = 0; x < 8; x++) { nz[x] = (b0 + x < fb.num_bins) ? (hist[b0 + x] != 0) : 0u; mine += nz[x]; }
    unsigned incl = wave_scan_incl(mine);
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    unsigned before = off + incl - mine;
    for (int x = 0; x < wid; x++) before += wsum[x];
#pragma unroll
    for (int x = 0; x < 8; x++) { if (b0 + x < fb.num_bins) eidx[(size_t)t * B + b0 + x] = before; before += nz[x]; }
}

// AddElement with concept drift, per slot, in stream order:  if A < w/decayWeight { w = A; min = bin }
// (histosketch.go:139-153).  Not a minimum: w may move either way, so elements are taken in order,
// but a wave tile whose fp32 minimum is not below the current threshold (with the fp32 band) cannot
// contain a trigger and is skipped; tiles that may are evaluated in fp64 and replayed exactly.
__global__ __launch_bounds__(256) void k_cws_resolve_drift(const double *__restrict__ rcb,
                                                           const double *__restrict__ f64,
                                                           const float *__restrict__ tilemin,
                                                           unsigned long long *__restrict__ mins,
                                                           double *__restrict__ weights, int slots,
                                                           int slot_begin, int ntiles, double decay_weight,
                                                           const DevState *st, FlushBatch fb) {
    extern __shared__ __align__(16) unsigned char smem[];
    float *tm = (float *)smem;                       // [wtiles]
    __shared__ int redi[4];
    __shared__ double s_w;
    __shared__ int s_first;
    const int slot = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wtiles = ntiles * 4, ngroups = (slots + SCAN_ROWS - 1) / SCAN_ROWS;
    const int32_t num_bins = fb.num_bins;
    const int gs = slot_begin + slot;
    const double *row = rcb + (size_t)slot * (size_t)num_bins * 3;
    double w = weights[gs];
    unsigned long long wm = mins[gs];
    const uint32_t gomask = batch_gomask(st, fb);

    for (int t = 0; t < (int)fb.count; t++) {
        if (!((gomask >> t) & 1u)) continue;
        __syncthreads();
        const float *tmin_t = tilemin + (((size_t)t * ngroups + slot / SCAN_ROWS) * wtiles) * SCAN_ROWS + (slot % SCAN_ROWS);
        const double *ft = f64 + (size_t)t * (size_t)num_bins;
        for (int x = tid; x < wtiles; x += blockDim.x) tm[x] = tmin_t[(size_t)x * SCAN_ROWS];
        __syncthreads();
        int from = 0;                                 // first tile not yet passed
        for (;;) {
            // first tile >= from that may hold a trigger for the current threshold
            const double thr = w / decay_weight;      // curMin of the reference (IEEE: +-Inf / NaN when weight is 0)
            int first = 0x7fffffff;
            if (!(thr != thr)) {                       // NaN threshold: nothing compares below it
                // fp32 screen; the band keeps it conservative (fp32 error of K*rcp is ~2e-7 relative)
                const double lim = thr + 1e-5 * fabs(thr) + 1e-37;
                for (int x = from + tid; x < wtiles; x += blockDim.x)
                    if ((double)tm[x] <= lim || !(lim < INFINITY)) { first = x; break; }
            }
            for (int off = 32; off; off >>= 1) { const int o = __shfl_xor(first, off); first = o < first ? o : first; }
            if (lane == 0) redi[wid] = first;
            __syncthreads();
            first = min(min(redi[0], redi[1]), min(redi[2], redi[3]));
            __syncthreads();
            if (first == 0x7fffffff) break;           // block-uniform
            // exact replay of tile `first`: one element per thread, in bin order
            const int32_t bin = first * WTILE + tid;
            double A = INFINITY; bool elem = false;
            if (bin < num_bins) {
                const double f = ft[bin];
                if (f != 0.0) {
                    const double r = row[(size_t)bin * 3 + 0], c = row[(size_t)bin * 3 + 1], b = row[(size_t)bin * 3 + 2];
                    const double Yka = exp(log(f) - b);
                    A = c / (Yka * exp(r));
                    elem = true;
                }
            }
            int pos = 0;                              // next position of the tile to look at
            for (;;) {
                const double th = w / decay_weight;
                int hit = 0x7fffffff;
                if (elem && tid >= pos && A < th) hit = tid;
                for (int off = 32; off; off >>= 1) { const int o = __shfl_xor(hit, off); hit = o < hit ? o : hit; }
                if (lane == 0) redi[wid] = hit;
                __syncthreads();
                hit = min(min(redi[0], redi[1]), min(redi[2], redi[3]));
                if (hit != 0x7fffffff && tid == hit) { s_w = A; s_first = bin; }
                __syncthreads();
                if (hit == 0x7fffffff) break;
                w = s_w; wm = (unsigned long long)s_first;
                pos = hit + 1;
            }
            from = first + 1;
        }
    }
    if (tid == 0) { weights[gs] = w; mins[gs] = wm; }
}

// ==========================================================================================
// newCWS on the device (histosketch.go:95-126).  The host walks Go's math/rand stream and hands
// over the (u1, u2) raw values of every Cheng attempt (cws_gen.h); here all attempts of a chunk are
// evaluated in parallel and the accepted gamma variates are compacted IN ORDER into the table:
// gamma #n -> entry n/2 = slot*B + bin, r if n is even, c = ln(gamma) if n is odd.
// ==========================================================================================
// ---- Go math/rand's additive lagged-Fibonacci stream, generated on the device -------------------
// y[m] = y[m-607] + y[m-273] (mod 2^64).  The stream itself does not depend on how the gamma sampler consumes it,
// so it is produced in chunks of 2^GO_RNG_JUMP_LOG2 values: k_alfg_jump walks the chunk start states with the
// jump polynomial (go_rng_jump.h: y[n + C + j] = sum_i coef[i] * y[n + i + j]), k_alfg_fill expands every chunk in
// parallel, 256 values per step (the shorter lag is 273).  windows[c] = the 607 values that end where chunk c begins.
__global__ __launch_bounds__(640) void k_alfg_jump(const uint64_t *__restrict__ coef, uint64_t *__restrict__ windows,
                                                   uint32_t first_chunk, uint32_t n_chunks) {
    __shared__ uint64_t E[1216], C[608];
    const int tid = threadIdx.x;
    if (tid < 607) { C[tid] = coef[tid]; E[tid] = windows[(size_t)first_chunk * 607 + tid]; }
    __syncthreads();
    for (uint32_t c = first_chunk; c + 1 < first_chunk + n_chunks; c++) {
        for (int base = 607; base < 1213; base += 273) {          // extend the window by 606 values, 273 at a time
            const int j = base + tid;
            if (tid < 273 && j < 1213) E[j] = E[j - 607] + E[j - 273];
            __syncthreads();
        }
        uint64_t acc = 0;
        if (tid < 607) for (int i = 0; i < 607; i++) acc += C[i] * E[i + tid];
        __syncthreads();
        if (tid < 607) { E[tid] = acc; windows[(size_t)(c + 1) * 607 + tid] = acc; }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_alfg_fill(const uint64_t *__restrict__ windows, uint64_t *__restrict__ raw,
                                                   uint32_t first_chunk, uint64_t chunk_len) {
    __shared__ uint64_t ring[1024];                               // the last 607 values live in a ring of 1024
    const int tid = threadIdx.x;
    const uint32_t c = first_chunk + blockIdx.x;
    const uint64_t *w = windows + (size_t)c * 607;
    for (int i = tid; i < 607; i += 256) ring[(1024 - 607 + i) & 1023] = w[i];   // window value i sits at position i - 607
    __syncthreads();
    uint64_t *out = raw + (size_t)c * chunk_len;
    for (uint64_t n = 0; n < chunk_len; n += 256) {
        const uint32_t at = (uint32_t)(n + tid);
        const uint64_t v = ring[(at - 607u) & 1023u] + ring[(at - 273u) & 1023u];
        __syncthreads();                                          // every read of this step before any write
        ring[at & 1023u] = v;
        out[n + tid] = v;
        __syncthreads();
    }
}
// positions of the stream whose value fails the gamma sampler's u1 range test or would make Float64() resample
// (type 1): rare (2e-7 / 2^-54 per value), resolved on the host into the `ev` list of k_cws_eval
__global__ __launch_bounds__(256) void k_rng_candidates(const uint64_t *__restrict__ raw, uint64_t n,
                                                        uint64_t *__restrict__ list, uint32_t cap,
                                                        unsigned int *__restrict__ count) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t x = raw[i] & 0x7fffffffffffffffull;
        const double u = (double)(long long)x * 0x1p-63;
        const bool skip = u == 1.0, fail = !(1e-7 < u && u < .9999999);
        if (skip || fail) {
            const unsigned at = atomicAdd(count, 1u);
            if (at < cap) list[at] = (i << 1) | (skip ? 1ull : 0ull);
        }
    }
}

constexpr int CWS_BLOCK = 1024;     // attempts per block (256 threads x 4)

// `pairs` is either the host-prepared (u1, u2) list of this chunk (raw == nullptr) or, in raw mode, unused: attempt
// g = first_attempt + i then reads the device-resident math/rand stream at raw[2g + d], raw[2g + d + 1], where d is
// the number of earlier attempts that died on the u1 range test (each consumed ONE value; `ev` holds, sorted, the
// index of the valid attempt that followed each of them).
__global__ __launch_bounds__(256) void k_cws_eval(const uint64_t *__restrict__ pairs, uint64_t n_attempts,
                                                  double *__restrict__ val, uint32_t *__restrict__ blkcnt,
                                                  double ainv, double bbb, double ccc, double magic,
                                                  const uint64_t *__restrict__ raw, uint64_t first_attempt,
                                                  const uint64_t *__restrict__ ev, uint32_t n_ev) {
    __shared__ unsigned red[4];
    unsigned cnt = 0;
#pragma unroll
    for (int x = 0; x < CWS_BLOCK / 256; x++) {
        const uint64_t i = (uint64_t)blockIdx.x * CWS_BLOCK + (uint64_t)x * 256 + threadIdx.x;
        double out = -1.0;                                       // < 0 marks a rejected attempt
        if (i < n_attempts) {
            uint64_t p0, p1;
            if (raw) {
                const uint64_t g = first_attempt + i;
                uint32_t lo = 0, hi = n_ev;                      // upper_bound(ev, g)
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (ev[mid] <= g) lo = mid + 1; else hi = mid; }
                const uint64_t a = 2 * g + lo;
                p0 = raw[a] & 0x7fffffffffffffffull; p1 = raw[a + 1] & 0x7fffffffffffffffull;
            } else { p0 = pairs[2 * i]; p1 = pairs[2 * i + 1]; }
            const double u1 = (double)(long long)p0 * 0x1p-63;
            const double u2 = 1.0 - (double)(long long)p1 * 0x1p-63;
            const double v = log(u1 / (1.0 - u1)) / ainv;
            const double xx = 2.0 * exp(v);
            const double z = u1 * u1 * u2;
            const double r = bbb + ccc * v - xx;
            if (r + magic - 4.5 * z >= 0.0 || r >= log(z)) { out = xx * 1.0; cnt++; }
            val[i] = out;
        }
    }
    for (int off = 32; off; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (lane_id() == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) blkcnt[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// exclusive scan of the block counts (single workgroup), advancing the running gamma count
__global__ __launch_bounds__(1024) void k_cws_scan_blocks(uint32_t *__restrict__ blkcnt, uint32_t nblk,
                                                          unsigned long long *__restrict__ gamma_total,
                                                          unsigned long long *__restrict__ chunk_base) {
    __shared__ uint32_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t per = (nblk + 1023) / 1024;
    const uint32_t lo = (uint32_t)tid * per, hi = lo + per < nblk ? lo + per : nblk;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += blkcnt[i];
    const uint32_t incl = wave_scan_incl(sum);
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    uint32_t before = incl - sum;
    for (int x = 0; x < wid; x++) before += wsum[x];
    for (uint32_t i = lo; i < hi; i++) { const uint32_t c = blkcnt[i]; blkcnt[i] = before; before += c; }
    if (tid == 1023) { *chunk_base = *gamma_total; *gamma_total += before; }
}

__global__ __launch_bounds__(256) void k_cws_scatter(const double *__restrict__ val, uint64_t n_attempts,
                                                     const uint32_t *__restrict__ blkoff,
                                                     const unsigned long long *__restrict__ chunk_base,
                                                     double *__restrict__ rcb, uint64_t num_bins,
                                                     uint64_t slot_begin, uint64_t slots, uint64_t sketch_size) {
    __shared__ unsigned wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint64_t i0 = (uint64_t)blockIdx.x * CWS_BLOCK + (uint64_t)tid * 4;   // 4 consecutive attempts per thread
    double v[4]; unsigned mine = 0;
#pragma unroll
    for (int x = 0; x < 4; x++) { v[x] = (i0 + x < n_attempts) ? val[i0 + x] : -1.0; mine += v[x] >= 0.0; }
    const unsigned incl = wave_scan_incl(mine);
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    unsigned long long n = *chunk_base + blkoff[blockIdx.x] + (incl - mine);
    for (int x = 0; x < wid; x++) n += wsum[x];
#pragma unroll
    for (int x = 0; x < 4; x++) {
        if (v[x] >= 0.0) {
            const unsigned long long entry = n >> 1;                  // slot * B + bin
            const unsigned long long slot = entry / num_bins;
            if (slot >= slot_begin && slot < slot_begin + slots && slot < sketch_size) {
                const unsigned long long at = (entry - slot_begin * num_bins) * 3 + (n & 1ull);
                rcb[at] = (n & 1ull) ? log(v[x]) : v[x];              // r = Gamma(2,1); c = ln(Gamma(2,1))
            }
            n++;
        }
    }
}

// b = U(0,1) * r with the separate uniform generator: entry i uses its i-th Float64
__global__ __launch_bounds__(256) void k_cws_beta(const uint64_t *__restrict__ uraw, uint64_t first_entry,
                                                  uint64_t n, double *__restrict__ rcb, uint64_t num_bins,
                                                  uint64_t slot_begin, uint64_t slots) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t entry = first_entry + i, slot = entry / num_bins;
        if (slot >= slot_begin && slot < slot_begin + slots) {
            const uint64_t at = (entry - slot_begin * num_bins) * 3;
            const double u = 0.0 + (double)(long long)(uraw[i] & 0x7fffffffffffffffull) * 0x1p-63 * (1.0 - 0.0);   // Float64Range(0, 1)
            rcb[at + 2] = u * rcb[at];
        }
    }
}

// ==========================================================================================
// hulk smash: pairwise distance matrix over N sketches of S slots.
// distances.GetDistance "jaccard" (distances.go:19-26) and GetWJD (distances.go:44-72) with the
// reference's quirk that BOTH weight vectors come from the subject sketch (sketchio.go:293-301).
// Thread (s, q) accumulates over the slots IN ORDER, so the fp64 sums are bit-identical to the Go
// loops; a 16x16 tile of pairs shares the slot chunks of its 16 subjects / 16 queries through LDS.
// ==========================================================================================
constexpr int SMASH_T = 16, SMASH_CH = 64;
__global__ __launch_bounds__(256) void k_smash(const unsigned long long *__restrict__ mins,
                                               const double *__restrict__ weights, uint32_t N, uint32_t S,
                                               int metric, double *__restrict__ out) {
    __shared__ unsigned long long ma[SMASH_T][SMASH_CH + 1], mb[SMASH_T][SMASH_CH + 1];
    __shared__ double wa[SMASH_T][SMASH_CH + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;        // query, subject inside the tile
    const uint32_t s = blockIdx.y * SMASH_T + ty, q = blockIdx.x * SMASH_T + tx;
    double intersect = 0.0, uni = 0.0;
    for (uint32_t c0 = 0; c0 < S; c0 += SMASH_CH) {
        for (int i = threadIdx.x; i < SMASH_T * SMASH_CH; i += 256) {
            const int r = i / SMASH_CH, c = i % SMASH_CH;
            const uint32_t sa = blockIdx.y * SMASH_T + r, qb = blockIdx.x * SMASH_T + r, col = c0 + c;
            const bool okc = col < S;
            ma[r][c] = (okc && sa < N) ? mins[(size_t)sa * S + col] : 0ull;
            wa[r][c] = (okc && sa < N) ? weights[(size_t)sa * S + col] : 0.0;
            mb[r][c] = (okc && qb < N) ? mins[(size_t)qb * S + col] : 0ull;
        }
        __syncthreads();
        const uint32_t lim = S - c0 < (uint32_t)SMASH_CH ? S - c0 : (uint32_t)SMASH_CH;
        if (metric == 1) {
            for (uint32_t c = 0; c < lim; c++) {
                // math.Max(math.Max(w,0), math.Max(-w,0)) == |w| (NaN stays NaN); weightB == weightA
                const double wgt = fabs(wa[ty][c]);
                if ((double)ma[ty][c] == (double)mb[tx][c]) { intersect += wgt; uni += wgt; }
                else uni += wgt;
            }
        } else {
            for (uint32_t c = 0; c < lim; c++) if ((double)ma[ty][c] == (double)mb[tx][c]) intersect += 1.0;
        }
        __syncthreads();
    }
    if (s < N && q < N)
        out[(size_t)s * N + q] = metric == 1 ? 1 - (intersect / uni) : 1.0 - (intersect / (double)S);
}

// K = c * exp(b - r) in fp64, rounded once to fp32 (pad columns stay 0: 0 * NaN = NaN, ignored)
__global__ __launch_bounds__(256) void k_build_k32(const double *__restrict__ rcb,
                                                   float *__restrict__ k32, int32_t num_bins,
                                                   size_t row_stride) {
    const int slot = blockIdx.y;
    const double *row = rcb + (size_t)slot * (size_t)num_bins * 3;
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < row_stride;
         b += (size_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (b < (size_t)num_bins) {
            const double r = row[b * 3 + 0], c = row[b * 3 + 1], bb = row[b * 3 + 2];
            v = (float)(c * exp(bb - r));
        }
        k32[(size_t)slot * row_stride + b] = v;
    }
}

// self-test: RN(1/r) by Newton == IEEE division for every r in [1, 2^31]
__global__ __launch_bounds__(256) void k_selftest_rcp(unsigned long long *mismatches) {
    unsigned bad = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; r <= 0x80000000ull;
         r += (uint64_t)gridDim.x * blockDim.x) {
        const double a = rcp_exact_u31((uint32_t)r);
        const double b = 1.0 / (double)(uint32_t)r;
        bad += (a != b);
        bad += (quot31_exact((uint32_t)r) != 0x1p31 / (double)(uint32_t)r);
    }
    for (int off = 32; off; off >>= 1) bad += __shfl_xor(bad, off);
    if (lane_id() == 0 && bad) atomicAdd(mismatches, (unsigned long long)bad);
}

__global__ void k_fill_f32(float *p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void k_add_hist(uint32_t *hist, const uint32_t *add, int32_t n) {
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) hist[i] += add[i];
}

}  // namespace

// ---------------------------------------------------------------------------- host wrappers
size_t minimizer_lds_per_wave(uint32_t xcap, uint32_t tab_size) {
    size_t words = (size_t)xcap + (xcap + 63) / 64 + tab_size + 128 + 64;
    size_t pk = ((size_t)xcap + 32 + 3) / 4 + 16;          // packed bases + slack for 3-dword reads
    pk = (pk + 7) & ~(size_t)7;
    return words * 8 + pk;
}
size_t minimizer_lds_per_block(uint32_t xcap, uint32_t tab_size, int waves) {
    return 256 + (size_t)waves * minimizer_lds_per_wave(xcap, tab_size);
}

hipError_t launch_minimizer_bin(hipStream_t s, const uint8_t *d_bases, const uint64_t *d_offsets,
                                uint64_t n_reads, MinimizerParams P, int block_threads,
                                uint32_t *d_hist, DevState *d_state, unsigned long long *d_min_slots,
                                const uint32_t *d_read_list, const uint32_t *d_read_list_count,
                                uint32_t list_blocks) {
    if (n_reads == 0) return hipSuccess;
    const int waves = block_threads / 64;
    P.lds_per_wave = (uint32_t)minimizer_lds_per_wave(P.xcap, P.tab_size);
    const size_t lds = minimizer_lds_per_block(P.xcap, P.tab_size, waves);
    uint64_t blocks = (n_reads + (uint64_t)waves * 4 - 1) / ((uint64_t)waves * 4);
    if (blocks > MIN_SLOTS) blocks = MIN_SLOTS;
    if (blocks < 1) blocks = 1;
    if (d_read_list) blocks = list_blocks;            // size unknown on the host: small fixed grid
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k_minimizer_bin,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_minimizer_bin, dim3((unsigned)blocks), dim3(block_threads), lds, s, d_bases,
                       d_offsets, n_reads, P, d_hist, d_state, d_min_slots, d_read_list, d_read_list_count);
    return hipGetLastError();
}

size_t minimizer_fast_lds(uint32_t, bool pair) {
    return 2048 + 16 * (size_t)FAST_TAB * 8 + 16 * 20 * 4 + 4 * (size_t)(pair ? FAST_RAW_PAIR : FAST_RAW) + 16 * (size_t)FAST_CAND * 8;
}

hipError_t launch_minimizer_fast(hipStream_t s, const uint8_t *d_bases, const uint64_t *d_offsets,
                                 uint64_t n_reads, MinimizerParams P, const MinimizerList &ml,
                                 DevState *d_state, unsigned long long *d_min_slots, uint32_t *d_slow_list,
                                 uint32_t *d_slow_count) {
    if (n_reads == 0) return hipSuccess;
    const bool pair = P.pair != 0;
    const size_t lds = minimizer_fast_lds(P.w, pair);
    const uint64_t blocks = (n_reads + 4 * FAST_READS_PER_WAVE - 1) / (4 * FAST_READS_PER_WAVE);
    const dim3 g((unsigned)blocks), b(256);
    // k <= 27: 64-bit minima through v_min_f64 (see umin64); HULK_NO_FMIN keeps the integer compares (A/B aid)
    static const bool no_fmin = getenv("HULK_NO_FMIN") != nullptr;
    const bool fm = P.k <= 27 && !no_fmin;
#define HULK_LAUNCH_FAST3(WM, FMv, DBGv, WEQv, KCv, PAIRv)                                                           \
    hipLaunchKernelGGL((k_minimizer_fast<WM, FMv, DBGv, WEQv, KCv, PAIRv>), g, b, lds, s, d_bases, d_offsets, n_reads, \
                       P, ml, d_state, d_min_slots, d_slow_list, d_slow_count)
#define HULK_LAUNCH_FAST2(WM, FMv, DBGv, WEQv, KCv)                                                                  \
    do { if (pair) HULK_LAUNCH_FAST3(WM, FMv, DBGv, WEQv, KCv, true); else HULK_LAUNCH_FAST3(WM, FMv, DBGv, WEQv, KCv, false); } while (0)
#define HULK_LAUNCH_FAST(WM)                                                                                         \
    do {                                                                                                             \
        const bool weq = P.w == WM;                                                                                  \
        if (P.debug) HULK_LAUNCH_FAST2(WM, false, true, false, 0);                                                   \
        else if (fm && weq && P.k == 21) HULK_LAUNCH_FAST2(WM, true, false, true, 21);   /* hulk's defaults: k=21, w=9 */ \
        else if (!fm && weq && P.k == 31) HULK_LAUNCH_FAST2(WM, false, false, true, 31); /* the largest k */          \
        else if (fm && weq) HULK_LAUNCH_FAST2(WM, true, false, true, 0);                                             \
        else if (fm) HULK_LAUNCH_FAST2(WM, true, false, false, 0);                                                   \
        else if (weq) HULK_LAUNCH_FAST2(WM, false, false, true, 0);                                                  \
        else HULK_LAUNCH_FAST2(WM, false, false, false, 0);                                                          \
    } while (0)
    if (P.w <= 4) HULK_LAUNCH_FAST(4);
    else if (P.w <= 9) HULK_LAUNCH_FAST(9);
    else HULK_LAUNCH_FAST(16);
#undef HULK_LAUNCH_FAST3
#undef HULK_LAUNCH_FAST2
#undef HULK_LAUNCH_FAST
    return hipGetLastError();
}

// K1b: jump hash of the list (dense key array); K1c: spectrum ranges in LDS, merged without atomics
hipError_t launch_minimizer_post(hipStream_t s, uint64_t n_reads, MinimizerParams P, const MinimizerList &ml,
                                 uint32_t *d_hists, uint32_t *d_zero_word) {
    if (n_reads == 0) return hipSuccess;
    hipError_t e = hipSuccess;
    const uint32_t n_regions = (uint32_t)((n_reads + FAST_READS_PER_WAVE - 1) / FAST_READS_PER_WAVE);
    const uint32_t nblk = (n_regions + 1023) / 1024;
    hipLaunchKernelGGL(k_region_bsum, dim3(nblk), dim3(1024), 0, s, ml.cnt, ml.bsum, n_regions);
    hipLaunchKernelGGL(k_region_offsets, dim3(nblk), dim3(1024), 0, s, ml.cnt, ml.bsum, ml.off, n_regions, ml.nib_over, d_zero_word);
    // k_jump_bin needs no LDS; a dummy allocation caps its occupancy so that the flush kernels of the
    // previous batch (other stream) find free wave slots next to it
    static int jump_lds = -1;
    if (jump_lds < 0) { const char *e = getenv("HULK_JUMP_LDS"); jump_lds = e ? atoi(e) : 0; }
    static int jump_c = -1;
    if (jump_c < 0) { const char *ec = getenv("HULK_JUMP_C"); jump_c = ec ? atoi(ec) : 0; }
    static int jump_cut = -1;
    if (jump_cut < 0) { const char *ec = getenv("HULK_JUMP_CUT"); jump_cut = ec ? atoi(ec) : 10; }
    const uint32_t cut = (jump_c || !ml.lo) ? 0u : (uint32_t)jump_cut;
    hipLaunchKernelGGL(k_jump_bin, dim3((n_regions + 3) / 4), dim3(256), (size_t)jump_lds, s, ml, n_regions, P.num_bins, jump_c, cut);
    if (cut) hipLaunchKernelGGL(k_jump_left, dim3((n_regions + 3) / 4), dim3(256), 0, s, ml, n_regions, P.num_bins);
    const uint32_t n_spectra = P.interval ? (uint32_t)((P.fill + n_reads + P.interval - 1) / P.interval) : 1u;
    const int nranges = (P.num_bins + HIST_RANGE - 1) / HIST_RANGE;
    static int parts_target = -1;
    // workgroups per launch (swept 256..768: 110-123 us for histogram + merge, flat)
    if (parts_target < 0) { const char *ep = getenv("HULK_HIST_BLOCKS"); parts_target = ep ? atoi(ep) : 512; }
    uint32_t n_parts = (uint32_t)parts_target / (uint32_t)(nranges * n_spectra);
    if (n_parts < 1) n_parts = 1;
    if (n_parts > ml.max_parts) n_parts = ml.max_parts;
    if (n_reads < 65536) n_parts = 1;
    static bool attr_set = false;
    if (!attr_set) {
        e = hipFuncSetAttribute((const void *)k_range_hist, hipFuncAttributeMaxDynamicSharedMemorySize, HIST_RANGE * 4);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    static int use_nib = -1;
    if (use_nib < 0) use_nib = getenv("HULK_NO_NIBBLE") ? 0 : 1;
    const uint32_t *only_if = nullptr;
    if (use_nib && ml.nib && ml.nib_over) {
        // ~131 k keys per part (16 parts per 100k-read interval, swept 49k..197k): a 4-bit counter then overflows only on
        // grossly repetitive input, which the exact kernels below pick up
        const int nr = (P.num_bins + NIB_BINS - 1) / NIB_BINS;
        const uint64_t rps = P.interval ? std::min<uint64_t>(P.interval, n_reads) : n_reads;
        static int keys_per_part = -1;
        if (keys_per_part < 0) { const char *ek = getenv("HULK_NIB_KEYS"); keys_per_part = ek ? atoi(ek) : 131072; }
        uint32_t np = (uint32_t)((rps * 20 + (uint64_t)keys_per_part - 1) / (uint64_t)keys_per_part);
        if (np < 1) np = 1;
        if (np > ml.nib_parts) np = ml.nib_parts;
        while (np > 1 && (uint64_t)np * n_spectra * nr > 2048) np--;
        const int words = ((std::min<int32_t>(P.num_bins, NIB_BINS) + 7) >> 3);
        static bool nib_attr = false;
        if (!nib_attr) {
            e = hipFuncSetAttribute((const void *)k_nibble_hist, hipFuncAttributeMaxDynamicSharedMemorySize, NIB_WORDS * 4);
            if (e != hipSuccess) return e;
            nib_attr = true;
        }
        const unsigned pg = (n_spectra * np + 7) / 8;
        hipLaunchKernelGGL(k_nibble_hist, dim3(8u * (unsigned)nr * pg), dim3(1024), (size_t)words * 4, s, ml, n_regions, P,
                           n_spectra, np, n_reads, nr);
        int nb = (nr * NIB_WORDS + 255) / 256; if (nb > 256) nb = 256;
        hipLaunchKernelGGL(k_nibble_merge, dim3(nb, n_spectra), dim3(256), 0, s, ml, d_hists, P, n_spectra, np, nr);
        only_if = ml.nib_over;                              // the exact kernels only recount flagged spectra
    }
    if (only_if) n_parts = 1;                                  // recount mode: rare, a small grid is enough
    const unsigned pair_groups = (n_spectra * n_parts + 7) / 8;
    hipLaunchKernelGGL(k_range_hist, dim3(8u * (unsigned)nranges * pair_groups), dim3(1024), HIST_RANGE * 4, s, ml, n_regions,
                       ml.partial, P, n_spectra, n_parts, n_reads, nranges, only_if, only_if ? d_hists : nullptr);
    if (!only_if) {
        int mb = (P.num_bins + 255) / 256; if (mb > 512) mb = 512;
        hipLaunchKernelGGL(k_merge_hist, dim3(mb, n_spectra), dim3(256), 0, s, ml.partial, d_hists, P, n_spectra, n_parts, only_if);
    }
    return hipGetLastError();
}

// a region holds at most one value per k-mer position of the wave's 16 reads
uint32_t minimizer_list_rcap(uint32_t w, bool pair) { return FAST_READS_PER_WAVE * (pair ? 2u * 16u * w - (w - 1u) : 16u * w); }

hipError_t launch_long_group(hipStream_t s, const uint8_t *d_bases, const LongSeqDesc *d_desc, uint32_t n_seqs,
                             uint64_t max_npos, MinimizerParams P, uint64_t *d_xs, uint8_t *d_valid, uint64_t *d_table,
                             uint64_t table_total, uint32_t *d_hists, unsigned long long *d_min_slots) {
    // blocks per sequence: enough for the longest of the group, bounded so that the grid stays ~2^17 blocks
    uint64_t bx = (max_npos + 256 * LONG_PPT - 1) / (256 * LONG_PPT);
    const uint64_t cap = std::max<uint64_t>(1, 131072 / n_seqs);
    if (bx > cap) bx = cap;
    if (bx > 8192) bx = 8192;
    hipLaunchKernelGGL(k_fill_u64, dim3(4096), dim3(256), 0, s, d_table, table_total, TAB_EMPTY);
    hipLaunchKernelGGL(k_long_hash, dim3((unsigned)bx, n_seqs), dim3(256), 0, s, d_bases, d_desc, P, d_xs, d_valid);
    hipLaunchKernelGGL(k_long_emit, dim3((unsigned)bx, n_seqs), dim3(256), 0, s, d_desc, d_xs, d_valid, P, d_table,
                       d_hists, d_min_slots);
    return hipGetLastError();
}

hipError_t launch_count_used(hipStream_t s, const uint32_t *d_hists, DevState *st, const FlushBatch &fb) {
    int blocks = (fb.num_bins + 2047) / 2048; if (blocks > 128) blocks = 128;
    hipLaunchKernelGGL(k_count_used, dim3(blocks, fb.count), dim3(256), 0, s, d_hists, st, fb);
    return hipGetLastError();
}

hipError_t launch_cms_binorder(hipStream_t s, uint32_t *d_hists, const uint16_t *d_pos16, const uint8_t *d_meta8,
                               unsigned long long *d_ctr, uint32_t *d_segsum, unsigned long long *d_base,
                               double *d_f64, float *d_rcp32, int depth, int width, size_t row_stride,
                               DevState *st, const FlushBatch &fb) {
    const int chunks = (fb.num_bins + 63) / 64;
    const int seg_chunks = (chunks + CMS_SEGS - 1) / CMS_SEGS;
    const size_t lds1 = (size_t)depth * width * 4;
    const size_t lds3 = (size_t)depth * width * 8 + (size_t)2 * depth * CMS_GROUP * 64 * 8;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)k_cms_freq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_cms_segsum, dim3(CMS_SEGS, fb.count), dim3(512), lds1, s, d_hists, d_pos16, d_segsum, depth, width,
                       seg_chunks, st, fb);
    hipLaunchKernelGGL(k_cms_base, dim3((depth * width + 255) / 256), dim3(256), 0, s, d_segsum, d_ctr, d_base, depth, width, st, fb);
    hipLaunchKernelGGL(k_cms_freq, dim3(CMS_SEGS, fb.count), dim3(512), lds3, s, d_hists, d_pos16, d_meta8, d_base, d_f64,
                       d_rcp32, depth, width, seg_chunks, row_stride, st, fb);
    return hipGetLastError();
}
size_t cms_binorder_entries(int depth, int width) { return (size_t)depth * CMS_SEGS * width; }

hipError_t launch_cmsd_binorder(hipStream_t s, uint32_t *d_hists, const uint16_t *d_pos16, const uint8_t *d_meta8,
                                const uint32_t *d_eidx, const uint32_t *d_etot, double *d_ctrd, double *d_segadd,
                                double *d_segfac, uint32_t *d_sege0, double *d_cstart, double *d_f64, float *d_rcp32,
                                int depth, int width, size_t row_stride, double omega, DevState *st, const FlushBatch &fb) {
    const int chunks = (fb.num_bins + 63) / 64;
    const int seg_chunks = (chunks + CMS_SEGS - 1) / CMS_SEGS;
    const size_t lds1 = (size_t)depth * width * 8;
    const size_t lds3 = (size_t)depth * width * 8 + (size_t)2 * depth * CMSD_GROUP * 64 * 8 + (size_t)depth * width * 2;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)k_cmsd_segsum, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_cmsd_freq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_cmsd_segsum, dim3(CMS_SEGS, fb.count), dim3(512), lds1, s, d_hists, d_pos16, d_eidx, d_etot, d_segadd,
                       d_segfac, d_sege0, depth, width, seg_chunks, omega, st, fb);
    hipLaunchKernelGGL(k_cmsd_base, dim3((depth * width + 255) / 256), dim3(256), 0, s, d_segadd, d_segfac, d_ctrd, d_cstart,
                       depth, width, st, fb);
    hipLaunchKernelGGL(k_cmsd_freq, dim3(CMS_SEGS, fb.count), dim3(512), lds3, s, d_hists, d_pos16, d_meta8, d_eidx, d_sege0,
                       d_cstart, d_f64, d_rcp32, depth, width, seg_chunks, row_stride, omega, st, fb);
    return hipGetLastError();
}

hipError_t launch_cws_scan(hipStream_t s, const float *d_k32, const float *d_rcp32, float *d_tilemin,
                           int slots, int ntiles, size_t row_stride, DevState *st, const FlushBatch &fb,
                           const float *d_kmin32, float *d_rext, const double *d_weights, int slot_begin,
                           unsigned long long *d_visited, double drift_dw) {
    const int groups = (slots + SCAN_ROWS - 1) / SCAN_ROWS;
    const int chunks = (ntiles + 7) / 8;
    if (d_kmin32)
        hipLaunchKernelGGL(k_rcp_extrema, dim3(ntiles, fb.count), dim3(256), 0, s, d_rcp32, d_rext, ntiles, row_stride, st, fb);
    hipLaunchKernelGGL(k_cws_scan, dim3((unsigned)(chunks * 8 * groups)), dim3(256), 0, s, d_k32, d_rcp32,
                       d_tilemin, slots, ntiles, row_stride, st, fb, d_kmin32, d_rext, d_weights, slot_begin, d_visited, drift_dw);
    return hipGetLastError();
}

hipError_t launch_slot_kmin(hipStream_t s, const float *d_kmin32, float *d_kminslot, int slots, int ntiles) {
    hipLaunchKernelGGL(k_slot_kmin, dim3(slots), dim3(256), 0, s, d_kmin32, d_kminslot, ntiles * 4);
    return hipGetLastError();
}

hipError_t launch_flush_decide(hipStream_t s, const unsigned long long *d_ctr, int ncounters, const float *d_kminslot,
                               const double *d_weights, int slots, int slot_begin, DevState *st, const FlushBatch &fb,
                               int enable) {
    hipLaunchKernelGGL(k_flush_decide, dim3(1), dim3(1024), 0, s, d_ctr, ncounters, d_kminslot, d_weights, slots,
                       slot_begin, st, fb, enable);
    return hipGetLastError();
}

hipError_t launch_tile_kmin(hipStream_t s, const float *d_k32, float *d_kmin32, int slots, int ntiles, size_t row_stride) {
    hipLaunchKernelGGL(k_tile_kmin, dim3(ntiles, slots), dim3(256), 0, s, d_k32, d_kmin32, ntiles, row_stride);
    return hipGetLastError();
}

hipError_t launch_cws_resolve(hipStream_t s, const double *d_rcb, const double *d_f64,
                              const float *d_tilemin, double *d_candA, int32_t *d_candB,
                              unsigned long long *d_mins, double *d_weights,
                              int slots, int slot_begin, int ntiles, DevState *st, const FlushBatch &fb) {
    hipLaunchKernelGGL(k_cws_resolve, dim3(slots, fb.count), dim3(256), (size_t)ntiles * 4 * sizeof(float), s,
                       d_rcb, d_f64, d_tilemin, d_candA, d_candB, slots, ntiles, st, fb);
    hipLaunchKernelGGL(k_cws_apply, dim3((slots + 255) / 256), dim3(256), 0, s, d_candA, d_candB, d_mins,
                       d_weights, slots, slot_begin, st, fb);
    return hipGetLastError();
}

hipError_t launch_elem_index(hipStream_t s, const uint32_t *d_hists, uint32_t *d_blkcnt, uint32_t *d_eidx,
                             uint32_t *d_etot, const FlushBatch &fb) {
    const int nblk = (fb.num_bins + EIDX_BLOCK - 1) / EIDX_BLOCK;
    hipLaunchKernelGGL(k_elem_count, dim3(nblk, fb.count), dim3(256), 0, s, d_hists, d_blkcnt, nblk, fb);
    hipLaunchKernelGGL(k_elem_index, dim3(nblk, fb.count), dim3(256), 0, s, d_hists, d_blkcnt, d_eidx, d_etot, nblk, fb);
    return hipGetLastError();
}
int elem_index_blocks(int32_t num_bins) { return (num_bins + EIDX_BLOCK - 1) / EIDX_BLOCK; }

hipError_t launch_cws_resolve_drift(hipStream_t s, const double *d_rcb, const double *d_f64,
                                    const float *d_tilemin, unsigned long long *d_mins, double *d_weights,
                                    int slots, int slot_begin, int ntiles, double decay_weight,
                                    DevState *st, const FlushBatch &fb) {
    hipLaunchKernelGGL(k_cws_resolve_drift, dim3(slots), dim3(256), (size_t)ntiles * 4 * sizeof(float), s,
                       d_rcb, d_f64, d_tilemin, d_mins, d_weights, slots, slot_begin, ntiles, decay_weight, st, fb);
    return hipGetLastError();
}

hipError_t launch_cws_chunk(hipStream_t s, const uint64_t *d_pairs, uint64_t n_attempts, double *d_val,
                            uint32_t *d_blkcnt, unsigned long long *d_gamma_total, unsigned long long *d_chunk_base,
                            double *d_rcb, uint64_t num_bins, uint64_t slot_begin, uint64_t slots,
                            uint64_t sketch_size, double ainv, double bbb, double ccc, double magic,
                            const uint64_t *d_raw, uint64_t first_attempt, const uint64_t *d_ev, uint32_t n_ev) {
    const uint32_t nblk = (uint32_t)((n_attempts + CWS_BLOCK - 1) / CWS_BLOCK);
    hipLaunchKernelGGL(k_cws_eval, dim3(nblk), dim3(256), 0, s, d_pairs, n_attempts, d_val, d_blkcnt, ainv, bbb, ccc, magic,
                       d_raw, first_attempt, d_ev, n_ev);
    hipLaunchKernelGGL(k_cws_scan_blocks, dim3(1), dim3(1024), 0, s, d_blkcnt, nblk, d_gamma_total, d_chunk_base);
    hipLaunchKernelGGL(k_cws_scatter, dim3(nblk), dim3(256), 0, s, d_val, n_attempts, d_blkcnt, d_chunk_base, d_rcb,
                       num_bins, slot_begin, slots, sketch_size);
    return hipGetLastError();
}

hipError_t launch_alfg(hipStream_t s, const uint64_t *d_coef, uint64_t *d_windows, uint64_t *d_raw, uint32_t first_chunk,
                       uint32_t n_chunks, uint64_t chunk_len) {
    // windows[first_chunk] is valid; produces windows[first_chunk + 1 .. first_chunk + n_chunks) and the chunks themselves
    hipLaunchKernelGGL(k_alfg_jump, dim3(1), dim3(640), 0, s, d_coef, d_windows, first_chunk, n_chunks);
    hipLaunchKernelGGL(k_alfg_fill, dim3(n_chunks), dim3(256), 0, s, d_windows, d_raw, first_chunk, chunk_len);
    return hipGetLastError();
}

hipError_t launch_rng_candidates(hipStream_t s, const uint64_t *d_raw, uint64_t n, uint64_t *d_list, uint32_t cap,
                                 unsigned int *d_count) {
    hipLaunchKernelGGL(k_rng_candidates, dim3(4096), dim3(256), 0, s, d_raw, n, d_list, cap, d_count);
    return hipGetLastError();
}

hipError_t launch_cws_beta(hipStream_t s, const uint64_t *d_uraw, uint64_t first_entry, uint64_t n, double *d_rcb,
                           uint64_t num_bins, uint64_t slot_begin, uint64_t slots) {
    hipLaunchKernelGGL(k_cws_beta, dim3(2048), dim3(256), 0, s, d_uraw, first_entry, n, d_rcb, num_bins, slot_begin, slots);
    return hipGetLastError();
}

hipError_t launch_smash(hipStream_t s, const unsigned long long *d_mins, const double *d_weights, uint32_t N, uint32_t S,
                        int metric, double *d_out) {
    if (N == 0) return hipSuccess;
    hipLaunchKernelGGL(k_smash, dim3((N + SMASH_T - 1) / SMASH_T, (N + SMASH_T - 1) / SMASH_T), dim3(256), 0, s, d_mins,
                       d_weights, N, S, metric, d_out);
    return hipGetLastError();
}

hipError_t launch_build_k32(hipStream_t s, const double *d_rcb, float *d_k32, int slots,
                            int32_t num_bins, size_t row_stride) {
    if (slots == 0) return hipSuccess;
    int bx = (int)((row_stride + 255) / 256); if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(k_build_k32, dim3(bx, slots), dim3(256), 0, s, d_rcb, d_k32, num_bins, row_stride);
    return hipGetLastError();
}

hipError_t launch_selftest_rcp(hipStream_t s, unsigned long long *d_mismatches) {
    hipLaunchKernelGGL(k_selftest_rcp, dim3(4096), dim3(256), 0, s, d_mismatches);
    return hipGetLastError();
}

hipError_t launch_fill_f32(hipStream_t s, float *p, size_t n, float v) {
    hipLaunchKernelGGL(k_fill_f32, dim3(256), dim3(256), 0, s, p, n, v);
    return hipGetLastError();
}

hipError_t launch_add_hist(hipStream_t s, uint32_t *d_hist, const uint32_t *d_add, int32_t num_bins) {
    int blocks = (num_bins + 255) / 256; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_add_hist, dim3(blocks), dim3(256), 0, s, d_hist, d_add, num_bins);
    return hipGetLastError();
}

}  // namespace hulk
