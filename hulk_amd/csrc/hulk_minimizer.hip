// hulk_minimizer.hip — reads -> distinct minimizers per read (reference: src/minimizer/minimizer.go:96-204).
//   K1a k_minimizer_fast   short reads: a 16-lane group per read, minimizer list in HBM (docs/EXPERIMENTS.md)
//       k_minimizer_bin    reads of up to 1024 k-mer positions, any bytes (fused jump hash + atomics)
//       k_long_hash/k_long_emit   long reads and contigs, grouped launches
// All kernels are wave64 code for CDNA4; none of them has a CPU or library fallback.
#include "hulk_device.h"

#include <type_traits>

#include <math.h>
#include <stdlib.h>
#include <algorithm>

namespace hulk {
namespace {

// ------------------------------------------------------------------------------------------
// K1: minimizers + binning.  One wave owns a read at a time.
//
// LDS per wave:  Xs[xcap] u64   hashed k-mer (or X_NONE) per k-mer position
//                vm[xcap/64] u64 validity masks (position not skipped)
//                tab[tab_size] u64  open-addressing set = the per-read golang-set
//                q[128] u64     distinct minimizers waiting for a full-wave jump-hash pass
//                pk[...] u8     2-bit packed bases, base p at bits 2(p%4) of byte p/4
//                pkn[...] u8    "code 4" flag of base p at bit 2(p%4) of byte p/4 (reads with N and the like)
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
    if (read_list) {                                // second pass over the reads the fast kernel deferred
        n_reads = *read_list_count;
        if ((uint64_t)blockIdx.x * (blockDim.x >> 6) >= n_reads) return;   // the grid is sized for a long list: most blocks have nothing to do
    }
    for (int t = threadIdx.x; t < 256; t += blockDim.x) lut[t] = nt4_of((unsigned)t);
    __syncthreads();

    const int lane = lane_id();
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
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
    const size_t pkbytes = ((((size_t)xcap + 32 + 3) / 4 + 16) + 7) & ~(size_t)7;
    uint8_t *pkn8 = pk8 + pkbytes;                             // "code 4" flags, same layout as pk (low bit of a base's pair)
    const uint32_t *pkn32 = (const uint32_t *)pkn8;

    const int32_t k = (int32_t)P.k, w = (int32_t)P.w;
    const int32_t wwin = w > 0 ? w : 1;   // w == 0: the deque is emptied every step, same window as w == 1
    const uint64_t mask = (1ull << (2 * k)) - 1;

    for (uint32_t s = lane; s < tabn; s += 64) tab[s] = TAB_EMPTY;
    wave_sync();

    uint32_t qn = 0;                 // wave-uniform
    unsigned long long nmin = 0;     // wave-uniform
    const uint32_t dbg = P.debug;    // ablation switches for tools/archive/k1_ablate.py (0 in production)
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
                // (an ALIGNED dword with at least one byte inside the buffer lies in the buffer's last page: loading it whole is
                //  safe, and a buffer need not end on a dword border — its last 1-3 bases are in such a dword)
                if (sh && al + 4 < (uintptr_t)bases + P.bases_bytes) hi = *(const uint32_t *)(al + 4);
                const uint32_t by = sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
                const int nv = left < 4 ? (int)left : 4;
                unsigned pack = 0, npack = 0;
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    unsigned c = lut[(by >> (8 * t)) & 0xff];
                    if (t < nv) { sawN |= (c > 3); pack |= (c & 3u) << (2 * t); npack |= (c >> 2) << (2 * t); }
                }
                pk8[p >> 2] = (uint8_t)pack;
                pkn8[p >> 2] = (uint8_t)npack;
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
                {
                    const uint32_t bo = 2u * (uint32_t)j, d = bo >> 5, o = bo & 31u;
                    const uint64_t lo = (uint64_t)pk32[d] | ((uint64_t)pk32[d + 1] << 32);
                    uint64_t W = o ? (lo >> o) | ((uint64_t)pk32[d + 2] << (64 - o)) : lo;
                    W &= mask;                    // base j at bits 0..1, base i at bits 2(k-1)..
                    r = (~W) & mask;
                    if (hasN) {
                        // The reference does not special-case code 4 (minimizer.go:118-122): `f = (f<<2 | c) & mask` ORs bit 2
                        // into the PREVIOUS base's pair, `r = r>>2 | (3^c) << shift` (r is never masked) puts 7 at the top —
                        // complement 3 for the base itself, and bit 2k, which one step later is the low bit of the NEXT
                        // base's pair.  So with N(p) = "base p has code 4" and code 0 stored for such a base:
                        //   f: pair of base p |= N(p+1) for every base of the k-mer but its last;
                        //   r: pair of base p |= N(p-1) for every base of the k-mer (p-1 may lie in front of it), and
                        //      bit 2k is set when the k-mer's last base is an N.
                        // X = the flags of bases j-1 .. j+k-1 at bits 0, 2, ..., 2k.
                        const uint32_t nbo = j > 0 ? bo - 2u : 0u, nd = nbo >> 5, no = nbo & 31u;
                        const uint64_t nlo = (uint64_t)pkn32[nd] | ((uint64_t)pkn32[nd + 1] << 32);
                        uint64_t X = no ? (nlo >> no) | ((uint64_t)pkn32[nd + 2] << (64 - no)) : nlo;
                        if (j == 0) X <<= 2;
                        W |= (X >> 4) & (mask >> 2);
                        r |= X & ((mask << 1) | 1ull);
                    }
                    const uint64_t rev = __brevll(W) >> (64 - 2 * k);
                    f = ((rev >> 1) & 0x5555555555555555ull) | ((rev & 0x5555555555555555ull) << 1);
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
    // same-address atomics serialise at ~12 ns each on this chip: every block adds to its own slot of
    // min_slots[] instead (an atomic all the same: launches of the two work lanes run side by side)
    __shared__ unsigned long long blk_nmin[4];
    if (lane == 0) blk_nmin[wid] = nmin;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int x = 0; x < nw; x++) t += blk_nmin[x];
        if (t) atomicAdd(&min_slots[blockIdx.x], t);
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
// <= 64 run starts.  Anything else is marked in the region's deferred mask and handled by k_minimizer_bin.
//
// LDS per group: tab[128] u64 | pk[20] u32 | pkn[20] u32 (code-4 flags)      per wave: raw ASCII of its 16 reads      per group: cs[64] u64
// ------------------------------------------------------------------------------------------
#ifndef HULK_FAST_TAB
#define HULK_FAST_TAB 128
#endif
#ifndef HULK_FAST_PAD
#define HULK_FAST_PAD 0
#endif
constexpr int FAST_TAB = HULK_FAST_TAB;       // set slots per 16-lane group (a pair of groups sharing a read uses both tables)
constexpr int FAST_PAD = HULK_FAST_PAD;
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
                                                        unsigned long long *__restrict__ min_slots) {
    extern __shared__ __align__(16) unsigned char smem[];
    // (the first 2 KB of LDS held ASCII -> 2-bit tables once; the layout behind them is unchanged)

    // KEY5 (the k = 31 instance): minimizer values use all 64 bits there (hash << 8 wraps), so they are not comparable as
    // doubles — but X = (hash mod 2^56) << 8 | span with 1 <= span <= 31 is order-isomorphic to the 61-bit key
    // (hash mod 2^56) << 5 | span, which is.  The window minima, run starts and the candidate list work on keys (one
    // v_min_f64 per minimum instead of v_cmp_lt_u64 + two v_cndmask); a key becomes its X again when it enters the set.
    constexpr bool KEY5 = !FM && WEQ && KC == 31;
    constexpr bool FMM = FM || KEY5;                               // 64-bit minima through v_min_f64
    constexpr uint64_t XN = FMM ? 0x7FF0000000000000ull : X_NONE;  // "no value": above every minimizer value / key
    // (wid is wave-uniform: keeping it in an SGPR also keeps it out of the register allocator's way — hipcc 7.2 lost the
    // VGPR copy across the main loop in the <16, false, false, false, 0, true> instance, see docs/EXPERIMENTS.md)
    const int lane = lane_id(), wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
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
    // NV: reads with `N` (code 4, minimizer.go:118-122) stay in this kernel.  The reference does not special-case code 4:
    // `f = (f<<2 | c) & mask` ORs bit 2 into the PREVIOUS base's pair, `r = r>>2 | (3^c) << shift` (never masked) puts 7 at the
    // top.  With code 0 stored for such a base and N(p) = "base p has code 4" this is closed form (derived for k_minimizer_bin,
    // which checks it against the literal recurrence): in f the pair of base p gets |= N(p+1) for every base of the k-mer but
    // its last, in r the pair of base p gets |= N(p-1) for every base of the k-mer, bit 2k of r is set when the k-mer's last
    // base is an N, and f == r has to be tested for odd k too.  An iteration of the wave takes this variant of phase A only
    // when one of its 4 reads has an N (wave-uniform branch): +13 instructions per k-mer position there, nothing elsewhere.
    // So do reads with any other code-4 byte (IUPAC letters, anything seq_nt4_table maps to 4); U / u is T; only the raw bytes
    // 0..3 (which the table maps to themselves) defer the read to k_minimizer_bin.  (PAIR: the second
    // group's first k-mer needs N(posoff - 1), a base only its partner staged: read from the partner's flag dwords.)
#ifdef HULK_ANALYSIS_NO_NV      // tools/valu_by_line.py: the static count of the main loop without the N variant = what a read without N executes
    constexpr bool NV = false;
#else
    constexpr bool NV = (DX && HP && 2 * (KC + WM) <= 64) || (!DX && WM <= 9);   // one-window form, or the rolling form (the
#endif
                                                                                   // 16-position instances are at 128 VGPRs already)
    const int half = PAIR ? (grp & 1) : 0;                         // which half of the read this group takes
    const int sub = PAIR ? ((grp & 3) >> 1) : (grp & 3);           // read of the iteration
    const int32_t posoff = half ? 16 * w - (w - 1) : 0;            // first k-mer position of this group
    uint64_t *tab = (uint64_t *)(smem + FAST_PAD) + (size_t)(PAIR ? (grp & ~1) : grp) * FAST_TAB;   // the per-read set
    uint32_t *pk32 = (uint32_t *)(smem + FAST_PAD + 16 * FAST_TAB * 8) + grp * 20;
    uint32_t *pkn32 = (uint32_t *)(smem + FAST_PAD + 16 * FAST_TAB * 8 + 16 * 20 * 4) + grp * 20;   // "code 4" flags, pk's layout
    uint32_t *raw32 = (uint32_t *)(smem + FAST_PAD + 16 * FAST_TAB * 8 + 2 * 16 * 20 * 4) + (size_t)wid * (RAWB / 4);
    uint64_t *cs = (uint64_t *)(smem + FAST_PAD + 16 * FAST_TAB * 8 + 2 * 16 * 20 * 4 + 4 * RAWB) + (size_t)grp * FAST_CAND;
    {   // every group empties its OWN table (a pair's set spans both of its groups' tables)
        uint64_t *own = (uint64_t *)(smem + FAST_PAD) + (size_t)grp * FAST_TAB;
#pragma unroll
        for (int x = 0; x < FAST_TAB / 16; x++) own[gl + 16 * x] = TAB_EMPTY;
    }
    __syncthreads();

    // this wave owns FAST_READS_PER_WAVE consecutive reads and one region of the minimizer list
    const uint64_t region = (uint64_t)blockIdx.x * 4 + (uint64_t)wid;
    const uint64_t wave_first = region * FAST_READS_PER_WAVE;
    uint64_t *xl = ml.x + region * ml.rcap;
    uint8_t *sl8 = ml.slot + region * ml.rcap;
    uint32_t wcount = 0;              // wave-uniform: values written to the region so far
    uint32_t dmask = 0;               // wave-uniform: reads of the region handed to the generic kernel
    // ablation switches (tools/archive/k1_ablate.py) exist only in the DBG instantiation: in the production kernel they
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
        const bool first_ok = raw_a0 < lim;
#pragma unroll
        for (int x = 0; x < NCH; x++) {
            const uintptr_t a = raw_a0 + 16u * (uint32_t)(lane + 64 * x);
            whole[x] = a < raw_end + 16 && a < lim;       // (an aligned 16-byte chunk with a byte inside the buffer lies in its last page)
            v[x] = first_ok ? *(const uint4 *)(whole[x] ? a : raw_a0) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int x = 0; x < NCH; x++) {
            const uintptr_t a = raw_a0 + 16u * (uint32_t)(lane + 64 * x);
            if (whole[x]) *(uint4 *)(raw32 + 4 * (lane + 64 * x)) = v[x];
            else if (a < raw_end + 16)                          // the chunk lies behind the buffer
                for (int y = 0; y < 4; y++) raw32[4 * (lane + 64 * x) + y] = 0u;
        }
    }
    wave_sync();

    // spectrum slot of the wave's first read: ONE 64-bit division per wave (it used to be a quarter of
    // all instructions when done per read); the following 15 reads step from it
    uint32_t slot0 = P.ring_base; uint64_t rem0 = 0;
    if (P.interval) {
        // a launch covers at most ring_n - 1 intervals (hulk_flush.hip), so the quotient is found by a short
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
        bool sawN = false, softN = false;                      // sawN: a byte this kernel cannot take; softN (NV): an `N`
        if (act && !defer) {
            const int32_t p = 16 * gl;
            uint32_t pack = 0, npack = 0;
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
                    for (int x = 0; x < 5; x++) d[x] = (al + 4 * x < end) ? *(const uint32_t *)(al + 4 * x) : 0u;
                }
                // ASCII -> 2-bit, four bases per dword, no table: fold the case, code = (c >> 1 ^ c >> 2) & 3
                // (A 0, C 1, G 2, T 3), and prove it by mapping the codes back to letters with one v_perm_b32:
                // any byte that does not come back (N, U, 0..3, anything else) is looked at below.
                // (c * 0x01041040) >> 24 gathers the four codes of a dword
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
                // the last read of the call: what follows it is not a read; only its own bytes count
                const bool lastread = Lg - p < 16 && lastwave && idx + 1 == nrd;
                if (bad != 0 || lastread) {                        // rare: look at the bytes one by one
                    const int nv = lastread ? (int)(Lg - p) : 16;
                    uint32_t hard = 0;
#pragma unroll
                    for (int x = 0; x < 4; x++) {
                        const uint32_t by = __builtin_amdgcn_alignbit(d[x + 1], d[x], sh);
                        const uint32_t up = by & 0xDFDFDFDFu, e = up >> 1, c = (e ^ (e >> 1)) & 0x03030303u;
                        const int nb = nv - 4 * x;
                        const uint32_t m = nb >= 4 ? ~0u : nb <= 0 ? 0u : (1u << (8 * nb)) - 1u;
                        const uint32_t mm = (__builtin_amdgcn_perm(0u, 0x54474341u, c) ^ up) & m;      // bytes that did not come back
                        if (NV) {
                            // 0x80 per byte that did not come back.  seq_nt4_table (minimizer.go:23-40) gives U / u the code of
                            // T — which the formula already produced — the raw bytes 0..3 their own value (not taken here: the
                            // read is deferred), and EVERY other byte code 4: N, the IUPAC letters, anything else.  Those get a
                            // flag, and code 0 is what is stored for them (the formula gives 0 for N / n only: cleared below).
                            const uint32_t nz = (((mm & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | mm) & 0x80808080u;
                            const uint32_t tU = up ^ 0x55555555u, t3 = by & 0xFCFCFCFCu;
                            const uint32_t isU = ~(((tU & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | tU | 0x7F7F7F7Fu);
                            const uint32_t is03 = ~(((t3 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t3 | 0x7F7F7F7Fu);
                            hard |= nz & is03;
                            const uint32_t b4 = (nz & ~isU & ~is03) >> 7;                  // bits 0, 8, 16, 24
                            const uint32_t fl = (b4 & 1u) | ((b4 >> 6) & 4u) | ((b4 >> 12) & 16u) | ((b4 >> 18) & 64u);   // -> bits 0, 2, 4, 6
                            npack |= fl << (8 * x);
                        } else hard |= mm;
                    }
                    sawN = hard != 0;
                    softN = NV && npack != 0;
                    if (NV) pack &= ~(npack * 3u);
                }
            }
            pk32[gl] = pack;
            if (gl < 4) pk32[16 + gl] = 0;                     // slack for 3-dword window reads
            if (NV) { pkn32[gl] = npack; if (gl < 4) pkn32[16 + gl] = 0; }
        }
        {
            // a byte the kernel cannot take anywhere in the read defers it as a whole (both groups of a pair must agree)
            const uint32_t gN = PAIR ? (uint32_t)(__ballot(sawN) >> (lane & 32)) : (uint32_t)(__ballot(sawN) >> gsh) & 0xffffu;
            if (gN) defer = true;
        }
        // NV: does any read of this iteration carry an N?  (wave-uniform; reads deferred for another reason do not count)
        const bool waveN = NV && __ballot(softN && !defer) != 0ull;
        wave_sync();
        if (dbg & 32u) { sink += pk32[gl]; wave_sync(); continue; }      // ablation: staging only

        // ---- phase A: rolling k-mers over the own block (registers)
        const int32_t p0 = gl * w;                              // first position of the lane's block, within the group
        const int32_t ap0 = posoff + p0;                        // ... within the read
        const bool mine = act && !defer && p0 < nposg;
        uint32_t validbits = 0;
        uint64_t X[WM];
#pragma unroll
        for (int t = 0; t < WM; t++) if (!WEQ || t >= w) X[t] = XN;   // (slots past the window size; none when w == WM)
        {
            // Positions past the read's end are always the END of a lane's block and of the read: their values only
            // reach windows that are not reported (validbits gates every report), so they are computed like any
            // other instead of being replaced by "no value" one by one.  The same holds for whole lanes: a lane without a
            // position of its own (behind the read's end, or of a read that is deferred or in error) hashes whatever its
            // staging dwords hold — values that only travel to lanes further behind, none of which reports — instead of
            // being masked out and given nine "no value" registers first (9 v_mov_b64 + the exec juggling per iteration).
            if (mine) validbits = (1u << (nposg - p0 < w ? nposg - p0 : w)) - 1u;
            const int32_t span0 = ap0 + k - 1 - w + 2;
            uint64_t f = 0, r = 0;
            uint32_t nb = 0, nn = 0;                            // next <=15 bases, 2 bits each; their code-4 flags (N variant)
            uint64_t Bb = 0, Cl = 0;
            // NV: Xn = the code-4 flags of bases p0-1, p0, p0+1, ... at bits 0, 2, 4, ... (p0 - 1 may be the partner group's)
            uint64_t Xn = 0, Hb = 0;
            if (NV && waveN) {
                const uint32_t bo = 2u * (uint32_t)p0, nbo = p0 > 0 ? bo - 2u : 0u, d = nbo >> 5, o = nbo & 31u;
                const uint64_t lo = (uint64_t)pkn32[d] | ((uint64_t)pkn32[d + 1] << 32);
                Xn = o ? (lo >> o) | ((uint64_t)pkn32[d + 2] << (64 - o)) : lo;
                if (p0 == 0) {
                    Xn <<= 2;
                    if (PAIR && half) {                              // base posoff - 1 lies in the partner group's part of the read
                        const uint32_t q = (uint32_t)(posoff - 1);
                        Xn |= (uint64_t)(((pkn32 - 20)[q >> 4] >> (2u * (q & 15u))) & 1u);
                    }
                }
            }
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
                    r = (~W) & mask;
                    if (NV && waveN) {
                        // the block's first k-mer in closed form (as k_minimizer_bin): f's pairs |= N(p+1) but the last base's,
                        // r's pairs |= N(p-1) and bit 2k = N(last base); the positions behind it roll the literal recurrence
                        const uint64_t X0 = (2 * k + 2 < 64) ? (Xn & ((1ull << (2 * k + 2)) - 1)) : Xn;
                        W |= (X0 >> 4) & (mask >> 2);
                        r |= X0 & ((mask << 1) | 1ull);
                    }
                    const uint64_t rev = __brevll(W) >> (64 - 2 * k);
                    f = ((rev >> 1) & 0x5555555555555555ull) | ((rev & 0x5555555555555555ull) << 1);
                }
                {
                    const uint32_t bo = 2u * (uint32_t)(p0 + k), d = bo >> 5, o = bo & 31u;
                    const uint64_t lo = (uint64_t)pk32[d] | ((uint64_t)pk32[d + 1] << 32);
                    nb = (uint32_t)(lo >> o);
                    if (NV && waveN) {
                        const uint64_t ln = (uint64_t)pkn32[d] | ((uint64_t)pkn32[d + 1] << 32);
                        nn = (uint32_t)(ln >> o);
                    }
                }
            }
            // NV, one-window form: Xn cut to the window (for r, bit 2k included); Hb = the flags of bases p0+1 .. in the forward
            // (first base on top) layout of Bb, one pair down (for f)
            if (NV && DX && waveN) {
                constexpr int NBW = KC + WM - 1;
                Xn &= (NBW + 1 < 32) ? ((1ull << (2 * (NBW + 1))) - 1) : ~0ull;
                const uint64_t Fn = (Xn >> 4) & ((1ull << (2 * (NBW - 1))) - 1);      // pair u: N(p0 + u + 1), u < NBW - 1
                const uint64_t rv = __brevll(Fn) >> (64 - 2 * NBW);
                Hb = ((rv >> 1) & 0x5555555555555555ull) | ((rv & 0x5555555555555555ull) << 1);
            }
            auto phaseA = [&](auto nvx_tag) {
            constexpr bool NVX = decltype(nvx_tag)::value;
#pragma unroll
            for (int t = 0; t < WM; t++) {
                if (DX) {
                    // dword form (v_alignbit + v_bfe) of f = (Bb >> 2(WM-1-t)) & mask, r = (Cl >> 2t) & mask
                    constexpr int HB = 2 * KC - 32;
                    const int sf = 2 * (WM - 1 - t), sr = 2 * t;
                    const uint32_t bl = (uint32_t)Bb, cl = (uint32_t)Cl;
                    uint32_t bh = (uint32_t)(Bb >> 32), ch = (uint32_t)(Cl >> 32);
                    asm("" : "+v"(bh), "+v"(ch));   // opaque: or hipcc re-fuses the dwords into 64-bit shifts + masks
                    uint32_t fl = sf ? __builtin_amdgcn_alignbit(bh, bl, sf) : bl, fh = __builtin_amdgcn_ubfe(bh, sf, HB);
                    uint32_t rl = sr ? __builtin_amdgcn_alignbit(ch, cl, sr) : cl, rh = __builtin_amdgcn_ubfe(ch, sr, HB);
                    if (NVX) {
                        // f: pair of base p |= N(p+1) for every base but the k-mer's last (the bottom pair);
                        // r: pair of base p |= N(p-1), and bit 2k = N(last base)
                        const uint32_t hl = (uint32_t)Hb, xl = (uint32_t)Xn;
                        uint32_t hh = (uint32_t)(Hb >> 32), xh = (uint32_t)(Xn >> 32);
                        asm("" : "+v"(hh), "+v"(xh));
                        fl |= (sf ? __builtin_amdgcn_alignbit(hh, hl, sf) : hl) & ~3u; fh |= __builtin_amdgcn_ubfe(hh, sf, HB);
                        rl |= sr ? __builtin_amdgcn_alignbit(xh, xl, sr) : xl; rh |= __builtin_amdgcn_ubfe(xh, sr, HB + 1);
                    }
                    f = ((uint64_t)fh << 32) | fl;
                    r = ((uint64_t)rh << 32) | rl;
                } else if (t) {
                    uint64_t c = nb & 3u; nb >>= 2;
                    if (NVX) { c |= (uint64_t)(nn & 1u) << 2; nn >>= 2; }   // code 4 as the reference rolls it (minimizer.go:118-122)
                    f = (f << 2 | c) & mask;
                    r = (r >> 2) | ((3ull ^ c) << shift);
                }
                if (t < w) {
                    // (a canonical k-mer has 2k <= 62 bits for every legal k: the f64 minimum is exact whatever FM says;
                    //  with an N as its last base r carries bit 2k: still far below 2^62, and then f is the minimum)
                    // (k = 31 with an N as the last base: r carries bit 62, and with A or N in the four bases before it its
                    //  bit pattern is a signalling NaN, which v_min_f64 does not pass through — integer minimum there)
                    const uint64_t canon = (NVX && k >= 31) ? (f < r ? f : r) : umin64<true>(f, r);
                    int32_t span = span0 + t;
                    if (span >= k) span = k;
                    uint64_t x;
                    if (HP && !(dbg & 16u)) x = hash64_pack_kc<HP ? KC : 21>(canon, (uint32_t)span);
                    else if (KEY5) x = (hash64(canon, mask) & 0x00FFFFFFFFFFFFFFull) << 5 | (uint64_t)(uint32_t)span;
                    else x = ((dbg & 16u) ? canon * 0x9E3779B97F4A7C15ull : hash64_fm<FM>(canon, mask)) << 8 | (uint64_t)(int64_t)span;
                    // f == r (a k-mer that is its own reverse complement: even k only — or any k once code-4 flags are in
                    // play) is skipped by the reference: the position neither reports nor takes part in a window
                    if ((NVX || !(k & 1)) && f == r) { x = XN; validbits &= ~(1u << t); }
                    X[t] = x;
                }
            }
            };
            if (NV && waveN) phaseA(std::true_type{}); else phaseA(std::false_type{});
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
                    if (t < w) h = umin64<FMM>(X[t], h);
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
                hp[t] = FMM ? (v | (xnfix & 0xffffffff00000000ull)) : (v | xnfix);
            }
            hp[WM - 1] = XN;
            const uint32_t pv = dpp_row_shr1(validbits);
            // m(pos) for the block, and for each position whether it continues the previous position's value
            uint32_t eqbits = 0;
            uint64_t g = XN, pm = whole;
#pragma unroll
            for (int t = 0; t < WM; t++) {
                g = umin64<FMM>(X[t], g);
                const uint64_t hpt = (t + 1 < w) ? hp[t] : XN;
                const uint64_t m = umin64<FMM>(hpt, g);
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
        {
            // deferred reads are marked in the region's mask (one store per wave at the end): a compact list is built by
            // k_region_offsets.  (An atomicAdd on ONE list counter per deferred read serialised at ~12 ns each: 5 % of
            // reads with an N doubled this kernel's time.)
            const uint64_t dbal = __ballot(act && defer && gl == 0 && half == 0);
            if (PAIR) dmask |= ((uint32_t)(dbal & 1u) | ((uint32_t)(dbal >> 32) & 1u) << 1) << (RPI * it);
            else dmask |= ((uint32_t)(dbal & 1u) | ((uint32_t)(dbal >> 16) & 1u) << 1 | ((uint32_t)(dbal >> 32) & 1u) << 2 |
                           ((uint32_t)(dbal >> 48) & 1u) << 3) << (RPI * it);
            if (act && defer) act = false;
        }
        wave_sync();

        // ---- per-read set + list append: one candidate per lane per round (<= 4 rounds).  A
        // candidate is new iff its compare-and-swap finds the slot empty; new values go straight to
        // the wave's region of the minimizer list (consecutive ranks = consecutive addresses).
        uint32_t myslot[FAST_CAND / 16];
        uint32_t newmask = 0;
        constexpr uint32_t TABM = (PAIR ? 2 * FAST_TAB : FAST_TAB) - 1;    // a pair owns two neighbouring tables
        // Two candidates per lane and pass (c and c + 16): both values are read and both compare-and-swaps issued before
        // either result is looked at — one LDS round trip per pair instead of two dependent ones, and half the ballots.
        // (Same set semantics: the LDS executes a wave's atomics in program order, so of two equal values the second finds
        // the first; a value that meets another in its slot probes on as before.)
#pragma unroll
        for (int pr = 0; pr < FAST_CAND / 32; pr++) {
            const uint32_t c0 = (uint32_t)gl + 32u * (uint32_t)pr, c1 = c0 + 16u;
            myslot[2 * pr] = 0; myslot[2 * pr + 1] = 0;
            if (!__any((int)(c0 < total))) break;
            const bool a0 = c0 < total, a1 = c1 < total;
            uint64_t x0 = 0, x1 = 0;
            if (a0) x0 = cs[c0];
            if (a1) x1 = cs[c1];
            if (KEY5) { x0 = (x0 & ~31ull) << 3 | (x0 & 31ull); x1 = (x1 & ~31ull) << 3 | (x1 & 31ull); }   // key -> X = hash << 8 | span
            uint32_t sl0 = ((uint32_t)(x0 >> 8) ^ (uint32_t)(x0 >> 37)) & TABM, sl1 = ((uint32_t)(x1 >> 8) ^ (uint32_t)(x1 >> 37)) & TABM;
            unsigned long long o0 = (unsigned long long)TAB_EMPTY, o1 = (unsigned long long)TAB_EMPTY;
            if (!(dbg & 4u)) {
                if (a0) o0 = atomicCAS((unsigned long long *)&tab[sl0], (unsigned long long)TAB_EMPTY, (unsigned long long)x0);
                if (a1) o1 = atomicCAS((unsigned long long *)&tab[sl1], (unsigned long long)TAB_EMPTY, (unsigned long long)x1);
                while (a0 && o0 != TAB_EMPTY && o0 != x0) {        // occupied by another value: probe on
                    sl0 = (sl0 + 1) & TABM;
                    o0 = atomicCAS((unsigned long long *)&tab[sl0], (unsigned long long)TAB_EMPTY, (unsigned long long)x0);
                }
                while (a1 && o1 != TAB_EMPTY && o1 != x1) {
                    sl1 = (sl1 + 1) & TABM;
                    o1 = atomicCAS((unsigned long long *)&tab[sl1], (unsigned long long)TAB_EMPTY, (unsigned long long)x1);
                }
            }
            const bool new0 = a0 && o0 == TAB_EMPTY, new1 = a1 && o1 == TAB_EMPTY;
            myslot[2 * pr] = sl0; myslot[2 * pr + 1] = sl1;
            const uint64_t nb0 = __ballot(new0), nb1 = __ballot(new1);
            if (nb0 | nb1) {
                const uint32_t n0 = (uint32_t)__popcll(nb0);
                if (new0) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(nb0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nb0, 0u));
                    newmask |= 1u << (2 * pr);
                    if (dbg & 1u) sink += (uint32_t)x0; else { xl[wcount + rank] = x0; sl8[wcount + rank] = (uint8_t)hslot; }
                }
                if (new1) {
                    const uint32_t rank = n0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(nb1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nb1, 0u));
                    newmask |= 2u << (2 * pr);
                    if (dbg & 1u) sink += (uint32_t)x1; else { xl[wcount + rank] = x1; sl8[wcount + rank] = (uint8_t)hslot; }
                }
                wcount += n0 + (uint32_t)__popcll(nb1);
            }
        }
        wave_sync();
        // empty the set again: only the slots this lane filled
#pragma unroll
        for (int rnd = 0; rnd < FAST_CAND / 16; rnd++)
            if ((newmask >> rnd) & 1u) tab[myslot[rnd]] = TAB_EMPTY;
        wave_sync();
    }
    if (lane == 0 && wave_first < n_reads) { ml.cnt[region] = wcount; ml.dmask[region] = dmask; }
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
// Long sequences (long reads, FASTA contigs: anything beyond the generic kernel's 1024 k-mer positions).
// A launch covers a GROUP of sequences (blockIdx.y = sequence of the group, blockIdx.x strides over its
// positions), each with its own slice of the scratch arrays and of the set table (LongSeqDesc), so
// 10-kb reads are not launch-bound.  k_long_hash: hashed canonical k-mer per position with the literal
// recurrence (N-safe); k_long_emit: windowed minimum per position; per-sequence set = open-addressing
// table in HBM (64-bit compare-and-swap), tried only where the window minimum differs from the one the
// previous position emitted (same set, ~5x fewer atomics); new values are jump-hashed and counted.
// ------------------------------------------------------------------------------------------
#ifndef HULK_LONG_PPT
#define HULK_LONG_PPT 8
#endif
constexpr int LONG_PPT = HULK_LONG_PPT;        // consecutive positions per thread in the long-sequence kernels
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
    // new values are queued per workgroup and jump-hashed together at the end, one value per lane: hashed where they
    // are found, a lane ran its ~13-step chain alone while the rest of the wave waited (~20 % of the lanes busy)
    constexpr unsigned QCAP = 2048;
    __shared__ uint64_t q[QCAP];
    __shared__ unsigned qn;
    if (threadIdx.x == 0) qn = 0;
    __syncthreads();
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
                if (P.debug & 2u) { fresh++; continue; }            // (ablation, profiling build: no set, no jump hash, no spectrum)
                // per-read set: only the thread whose compare-and-swap claims the slot counts the value
                uint64_t slot = (m ^ (m >> 29)) * 0x9E3779B97F4A7C15ull >> 20 & table_mask;
                for (;;) {
                    const unsigned long long old = atomicCAS((unsigned long long *)&table[slot], (unsigned long long)TAB_EMPTY,
                                                             (unsigned long long)m);
                    if (old == TAB_EMPTY) {
                        if (P.debug & 1u) { fresh++; break; }     // (ablation: the set only)
                        const unsigned at = atomicAdd(&qn, 1u);
                        if (at < QCAP) q[at] = m; else atomicAdd(&hist[jump_hash(m, P.num_bins)], 1u);   // (queue full: in place)
                        fresh++;
                        break;
                    }
                    if (old == m) break;
                    slot = (slot + 1) & table_mask;
                }
            }
            prev_emit = emit; mprev = m;
        }
    }
    __syncthreads();
    {
        const unsigned nq = qn < QCAP ? qn : QCAP;
        if (P.debug & 4u) { unsigned acc = 0; for (unsigned i = threadIdx.x; i < nq; i += blockDim.x) acc += (unsigned)jump_hash(q[i], P.num_bins); if (acc == 0xdeadbeefu) hist[0] = acc; }   // (ablation: jump hash without the spectrum's atomics)
        else
        for (unsigned i = threadIdx.x; i < nq; i += blockDim.x) atomicAdd(&hist[jump_hash(q[i], P.num_bins)], 1u);
    }
    for (int off = 32; off; off >>= 1) fresh += __shfl_xor(fresh, off);
    if (lane_id() == 0) red[threadIdx.x >> 6] = fresh;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = red[0] + red[1] + red[2] + red[3];
        if (t) atomicAdd(&min_slots[(blockIdx.x + 131u * blockIdx.y) & (MIN_SLOTS - 1)], (unsigned long long)t);
    }
}

// The two passes above as ONE tile kernel (round 6).  k_long_hash / k_long_emit hand 9 bytes per position through HBM and both touch
// them a thread at a time — 8 consecutive u64 per lane, 64 B between neighbouring lanes: every load instruction of a wave hit 64
// different lines; the window minimum alone (no set, no jump hash) took 11 of k_long_emit's 25 ms per Gbase, k_long_hash 6 more.
// Here a workgroup owns a TILE of a sequence: LONG_TILE = 2048 consecutive k-mer positions, of which the first H = w rounded up to
// 8 are context the tile in front of it reports (their hashes are recomputed: 0.8 % at w = 9).  The tile's bases are staged once, as
// seq_nt4_table codes (coalesced), every thread rolls the literal recurrence (N-safe, as k_long_hash) over its 8 positions into LDS,
// and the window minima, run starts, the per-sequence set (HBM, 64-bit CAS: unchanged) and the queue of new values follow from
// LDS.  Same values, same set, same counts as the two-pass form (HULK_LONG_TWO_PASS in the profiling build keeps it as comparator).
constexpr int LONG_TILE = 2048;                 // tile indices per workgroup = 256 threads x 8 positions
__global__ __launch_bounds__(256) void k_long_tile(const uint8_t *__restrict__ bases, const LongSeqDesc *__restrict__ desc,
                                                   MinimizerParams P, uint64_t *__restrict__ table_all,
                                                   uint32_t *__restrict__ hists, unsigned long long *__restrict__ min_slots) {
    constexpr unsigned QCAP = 2048;
    __shared__ uint8_t lut[256];
    __shared__ uint8_t code[LONG_TILE + 64];     // code[b] = nt4 of base (o - 1 + b)
    __shared__ uint64_t Xs[LONG_TILE];
    __shared__ uint8_t valid[LONG_TILE];
    __shared__ uint64_t q[QCAP];
    __shared__ unsigned qn, red[4];
    const int tid = threadIdx.x;
    lut[tid] = nt4_of((unsigned)tid);
    const LongSeqDesc d = desc[blockIdx.y];
    const uint8_t *seq = bases + d.seq_off;
    uint64_t *table = table_all + d.tab_off;
    const uint64_t table_mask = d.tab_mask;
    uint32_t *hist = hists + (size_t)d.hslot * (size_t)P.num_bins;
    const int64_t k = (int64_t)P.k, w = (int64_t)P.w, L = (int64_t)d.L;
    const int64_t wwin = w > 0 ? w : 1;
    const int64_t npos = L - k + 1;
    const int64_t H = (wwin + 7) & ~(int64_t)7;                  // context positions in front of a tile (>= w, a multiple of 8)
    const int64_t TP = LONG_TILE - H;                            // positions a tile reports
    const int64_t ntiles = (npos + TP - 1) / TP;
    const uint64_t mask = (1ull << (2 * k)) - 1, shift = (uint64_t)(2 * (k - 1));
    unsigned fresh = 0;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t o = tile * TP - H;                         // sequence position of tile index 0 (negative in the first tile)
        if (tid == 0) qn = 0;
        __syncthreads();                                         // (also: the previous tile's readers of code / Xs / q are done, lut is built)
        for (int b = tid; b < LONG_TILE + (int)k; b += 256) {
            const int64_t pos = o - 1 + b;
            code[b] = (pos >= 0 && pos < L) ? lut[seq[pos]] : (uint8_t)0;
        }
        __syncthreads();
        {   // hashed canonical k-mers of the thread's 8 positions (the literal recurrence: minimizer.go:114-160)
            const int i0 = tid * 8;
            uint64_t f = 0, r = 0;
            // base (first position) - 1 still reaches r's lowest pair when it is code 4: k_long_hash starts there too
            for (int b = i0; b < i0 + 8 + (int)k; b++) {
                const uint64_t c = code[b];
                f = (f << 2 | c) & mask;
                r = (r >> 2) | ((3ull ^ c) << shift);
                const int i = b - (int)k;                        // tile index of the k-mer that ends with this base
                if (i < i0) continue;
                const int64_t j = o + i, p = j + k - 1;
                uint64_t X = X_NONE; uint8_t ok = 0;
                if (j >= 0 && j < npos && f != r) {
                    const uint64_t canon = f > r ? r : f;
                    int64_t span = p - w + 2;
                    if (span >= k) span = k;
                    X = hash64(canon, mask) << 8 | (uint64_t)(int64_t)(int32_t)span;
                    ok = 1;
                }
                Xs[i] = X; valid[i] = ok;
            }
        }
        __syncthreads();
        if (tid * 8 >= (int)H) {
            const int i0 = tid * 8;
            // sliding window minimum m(i) = min Xs[i-w+1 .. i] (positions in front of the sequence hold "no value"); a full rescan
            // only when the value that leaves the window is the current minimum
            uint64_t m = X_NONE;
            for (int p = i0 - (int)wwin; p <= i0 - 1; p++) { const uint64_t x = Xs[p]; m = x < m ? x : m; }
            uint64_t mprev = m;
            bool prev_emit = valid[i0 - 1] && (o + i0 - 1) + k - 1 >= w - 1;
            for (int i = i0; i < i0 + 8; i++) {
                const int64_t j = o + i;
                if (j >= npos) break;
                const uint64_t x = Xs[i];
                if (Xs[i - (int)wwin] == m) {
                    m = x;
                    for (int p = i - (int)wwin + 1; p < i; p++) { const uint64_t y = Xs[p]; m = y < m ? y : m; }
                } else {
                    m = x < m ? x : m;
                }
                const bool emit = valid[i] && j + k - 1 >= w - 1;
                if (emit && !(prev_emit && mprev == m)) {
                    uint64_t slot = (m ^ (m >> 29)) * 0x9E3779B97F4A7C15ull >> 20 & table_mask;
                    for (;;) {
                        const unsigned long long old = atomicCAS((unsigned long long *)&table[slot], (unsigned long long)TAB_EMPTY,
                                                                 (unsigned long long)m);
                        if (old == TAB_EMPTY) {
                            const unsigned at = atomicAdd(&qn, 1u);
                            if (at < QCAP) q[at] = m; else atomicAdd(&hist[jump_hash(m, P.num_bins)], 1u);   // (queue full: in place)
                            fresh++;
                            break;
                        }
                        if (old == m) break;
                        slot = (slot + 1) & table_mask;
                    }
                }
                prev_emit = emit; mprev = m;
            }
        }
        __syncthreads();
        {
            const unsigned nq = qn < QCAP ? qn : QCAP;
            for (unsigned i = tid; i < nq; i += 256) atomicAdd(&hist[jump_hash(q[i], P.num_bins)], 1u);
        }
    }
    for (int off = 32; off; off >>= 1) fresh += __shfl_xor(fresh, off);
    if (lane_id() == 0) red[tid >> 6] = fresh;
    __syncthreads();
    if (tid == 0) {
        const unsigned t = red[0] + red[1] + red[2] + red[3];
        if (t) atomicAdd(&min_slots[(blockIdx.x + 131u * blockIdx.y) & (MIN_SLOTS - 1)], (unsigned long long)t);
    }
}

__global__ void k_fill_u64(uint64_t *p, uint64_t n, uint64_t v) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = v;
}

}  // namespace

// ---------------------------------------------------------------------------- host wrappers
size_t minimizer_lds_per_wave(uint32_t xcap, uint32_t tab_size) {
    size_t words = (size_t)xcap + (xcap + 63) / 64 + tab_size + 128 + 64;
    size_t pk = ((size_t)xcap + 32 + 3) / 4 + 16;          // packed bases + slack for 3-dword reads
    pk = (pk + 7) & ~(size_t)7;
    return words * 8 + 2 * pk;                              // ... and the code-4 flags in the same layout
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
    prof_mark(s, "k_minimizer_bin");
    hipLaunchKernelGGL(k_minimizer_bin, dim3((unsigned)blocks), dim3(block_threads), lds, s, d_bases,
                       d_offsets, n_reads, P, d_hist, d_state, d_min_slots, d_read_list, d_read_list_count);
    return hipGetLastError();
}

size_t minimizer_fast_lds(uint32_t, bool pair) {
    return FAST_PAD + 16 * (size_t)FAST_TAB * 8 + 2 * 16 * 20 * 4 + 4 * (size_t)(pair ? FAST_RAW_PAIR : FAST_RAW) + 16 * (size_t)FAST_CAND * 8;
}

hipError_t launch_minimizer_fast(hipStream_t s, const uint8_t *d_bases, const uint64_t *d_offsets,
                                 uint64_t n_reads, MinimizerParams P, const MinimizerList &ml,
                                 DevState *d_state, unsigned long long *d_min_slots) {
    if (n_reads == 0) return hipSuccess;
    const bool pair = P.pair != 0;
    size_t lds = minimizer_fast_lds(P.w, pair);
    {   // (profiling build) HULK_K1_LDS_TOTAL=bytes: the kernel's LDS request padded up to it — how it runs at 3 / 2 / 1 workgroups per CU
        static const char *e = HULK_EXP_ENV("HULK_K1_LDS_TOTAL");
        static const size_t want = e ? (size_t)atol(e) : 0;
        if (want > lds) {
            static bool once = false;
            if (!once && want > 65536) { once = true; (void)hipFuncSetAttribute((const void *)k_minimizer_fast<9, false, false, true, 31, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want); }
            lds = want;
        }
    }
    const uint64_t blocks = (n_reads + 4 * FAST_READS_PER_WAVE - 1) / (4 * FAST_READS_PER_WAVE);
    const dim3 g((unsigned)blocks), b(256);
    // k <= 27: 64-bit minima through v_min_f64 (see umin64); HULK_NO_FMIN keeps the integer compares (A/B aid)
    static const bool no_fmin = HULK_EXP_ENV("HULK_NO_FMIN") != nullptr;
    const bool fm = P.k <= 27 && !no_fmin;
    prof_mark(s, "k_minimizer_fast");
#define HULK_LAUNCH_FAST3(WM, FMv, DBGv, WEQv, KCv, PAIRv)                                                           \
    hipLaunchKernelGGL((k_minimizer_fast<WM, FMv, DBGv, WEQv, KCv, PAIRv>), g, b, lds, s, d_bases, d_offsets, n_reads, \
                       P, ml, d_state, d_min_slots)
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

// a region holds at most one value per k-mer position of the wave's 16 reads — and at most FAST_CAND per 16-lane group (a
// read with more run starts is deferred to the generic kernel and contributes none).  The second bound is the smaller one
// from w = 5 up (1024 + pad instead of 2304 entries at w = 9): the lists of a batch shrink from 3.4 to 1.6 GB per lane, of
// which the reads of C2 fill 40 % instead of 19 %.  The pad keeps the region stride off a power of two.
uint32_t minimizer_list_rcap(uint32_t w, bool pair) {
    static const uint32_t pad = [] { const char *e = HULK_EXP_ENV("HULK_RCAP_PAD"); return e ? (uint32_t)atol(e) : 64u; }();
    const uint32_t by_pos = FAST_READS_PER_WAVE * (pair ? 2u * 16u * w - (w - 1u) : 16u * w);
    const uint32_t by_cand = FAST_READS_PER_WAVE * (uint32_t)FAST_CAND * (pair ? 2u : 1u);
    return by_pos < by_cand + pad ? by_pos : by_cand + pad;
}

hipError_t launch_long_group(hipStream_t s, const uint8_t *d_bases, const LongSeqDesc *d_desc, uint32_t n_seqs,
                             uint64_t max_npos, MinimizerParams P, uint64_t *d_xs, uint8_t *d_valid, uint64_t *d_table,
                             uint64_t table_total, uint32_t *d_hists, unsigned long long *d_min_slots) {
    prof_mark(s, "k_fill_u64");
    hipLaunchKernelGGL(k_fill_u64, dim3(4096), dim3(256), 0, s, d_table, table_total, TAB_EMPTY);
    if (d_xs && d_valid) {                                        // the two-pass form (profiling build, HULK_LONG_TWO_PASS: the comparator)
        // blocks per sequence: enough for the longest of the group, bounded so that the grid stays ~2^17 blocks
        uint64_t bx = (max_npos + 256 * LONG_PPT - 1) / (256 * LONG_PPT);
        const uint64_t cap = std::max<uint64_t>(1, 131072 / n_seqs);
        if (bx > cap) bx = cap;
        if (bx > 8192) bx = 8192;
        prof_mark(s, "k_long_hash");
        hipLaunchKernelGGL(k_long_hash, dim3((unsigned)bx, n_seqs), dim3(256), 0, s, d_bases, d_desc, P, d_xs, d_valid);
        prof_mark(s, "k_long_emit");
        hipLaunchKernelGGL(k_long_emit, dim3((unsigned)bx, n_seqs), dim3(256), 0, s, d_desc, d_xs, d_valid, P, d_table,
                           d_hists, d_min_slots);
        return hipGetLastError();
    }
    // tiles per sequence: enough for the longest of the group (shorter ones leave their surplus workgroups at once), bounded so that
    // the grid stays ~2^18 workgroups: a workgroup then strides over its sequence's tiles
    const uint64_t H = (((uint64_t)(P.w ? P.w : 1) + 7) & ~7ull), TP = (uint64_t)LONG_TILE - H;
    uint64_t bx = (max_npos + TP - 1) / TP;
    const uint64_t cap = std::max<uint64_t>(1, 262144 / n_seqs);
    if (bx > cap) bx = cap;
    if (bx > 65535) bx = 65535;
    prof_mark(s, "k_long_tile");
    hipLaunchKernelGGL(k_long_tile, dim3((unsigned)bx, n_seqs), dim3(256), 0, s, d_bases, d_desc, P, d_table, d_hists, d_min_slots);
    return hipGetLastError();
}

}  // namespace hulk
