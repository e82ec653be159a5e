// hulk_spectrum.hip — minimizer list -> k-mer spectrum (reference: src/kmerspectrum/kmerspectrum.go:67-81, go-jump).
//   K1b k_jump_bin / k_jump_left   jump hash of the list (exact fp64 reciprocal, assembly step loop)
//   K1c k_nibble_hist/k_nibble_merge, k_range_hist/k_merge_hist   spectrum in LDS, no global atomics
#include "hulk_device.h"

#include <math.h>
#include <stdlib.h>
#include <algorithm>

namespace hulk {
namespace {

// ------------------------------------------------------------------------------------------
// K1b: jump hash of the minimizer list.  One wave per region; lanes take the region's values
// round-robin (lane l: l, l+64, ...) with the next value prefetched, so every lane stays busy
// with its own chain of ~ln(k^4) fp64 steps (no lock-step tail per 64 values); 8 waves/SIMD.
// Output: key = spectrum slot << 20 | bin  (k^4 < 2^20 for k <= 31).
// ------------------------------------------------------------------------------------------
// The step loop of k_jump_bin in assembly (fixed registers v40..v57, s60..s63): hipcc's version of the same loop
// carries two v_mov_b64 and a dozen scalar mask instructions per pair of steps; this one is the 17 VALU
// instructions of a step (16 since r * 2^-31 = ((key >> 33) + 1) * 2^-31 is ONE fma(float64(key >> 33), 2^-31, 2^-31) instead of an
// integer add before the conversion and an exponent adjustment after it) plus v_cmp / s_and and the exit test, the LCG state ping-ponging between v[40:41] and
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
    "v_cvt_f64_u32 v[46:47], v54\n\t"                                                \
    "v_fma_f64 v[46:47], v[46:47], %[p31], %[p31]\n\t"                               \
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
          [alo] "s"(0x87B0B0FDu), [ahi] "s"(0x27BB2EE6u), [cut] "s"(cut), [p31] "s"(0x3E00000000000000ull /* 2^-31 */)
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

// exclusive prefix sum of the region counts: per-block sums, then one block per 1024 regions.  The same two kernels
// scan the number of deferred reads per region (popcount of dmask) and write the compact deferred-read list the
// generic kernel works through, in read order.
__global__ __launch_bounds__(1024) void k_region_bsum(const uint32_t *__restrict__ cnt, uint32_t *__restrict__ bsum,
                                                      const uint32_t *__restrict__ dmask, uint32_t *__restrict__ dsum,
                                                      uint32_t n_regions) {
    __shared__ uint32_t wsum[16], wdsum[16];
    const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
    uint32_t v = i < n_regions ? cnt[i] : 0u;
    uint32_t dv = i < n_regions ? (uint32_t)__popc(dmask[i]) : 0u;
    for (int off = 32; off; off >>= 1) { v += __shfl_xor(v, off); dv += __shfl_xor(dv, off); }
    if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = v; wdsum[threadIdx.x >> 6] = dv; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0, d = 0;
        for (int x = 0; x < 16; x++) { t += wsum[x]; d += wdsum[x]; }
        bsum[blockIdx.x] = t; dsum[blockIdx.x] = d;
    }
}
__global__ __launch_bounds__(1024) void k_region_offsets(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ bsum,
                                                         uint32_t *__restrict__ off, uint32_t n_regions,
                                                         uint32_t *__restrict__ nib_over, const uint32_t *__restrict__ dmask,
                                                         const uint32_t *__restrict__ dsum, uint32_t *__restrict__ slow_list,
                                                         uint32_t *__restrict__ slow_count) {
    if (nib_over && blockIdx.x == 0 && threadIdx.x < RING_MAX) nib_over[threadIdx.x] = 0;
    __shared__ uint32_t wsum[16], wdsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t i = blockIdx.x * 1024u + (uint32_t)tid;
    uint32_t before = 0, dbefore = 0;
    for (uint32_t b = 0; b < blockIdx.x; b++) { before += bsum[b]; dbefore += dsum[b]; }   // same address for the whole block: broadcast
    const uint32_t v = i < n_regions ? cnt[i] : 0u;
    uint32_t m = i < n_regions ? dmask[i] : 0u;
    const uint32_t dv = (uint32_t)__popc(m);
    const uint32_t incl = wave_scan_incl(v), dincl = wave_scan_incl(dv);
    if (lane == 63) { wsum[wid] = incl; wdsum[wid] = dincl; }
    __syncthreads();
    for (int x = 0; x < wid; x++) { before += wsum[x]; dbefore += wdsum[x]; }
    if (i < n_regions) off[i] = before + incl - v;
    if (i + 1 == n_regions) { off[n_regions] = before + incl; *slow_count = dbefore + dincl; }
    uint32_t at = dbefore + dincl - dv;
    while (m) {                                                   // rare: the reads of this region the fast kernel deferred
        const uint32_t b = (uint32_t)__ffs((int)m) - 1u;
        slow_list[at++] = i * FAST_READS_PER_WAVE + b;
        m &= m - 1u;
    }
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
#ifndef HULK_NIB_RLOG_DEFAULT
#define HULK_NIB_RLOG_DEFAULT 18
#endif
#ifndef HULK_NIB_BLOCK_DEFAULT
#define HULK_NIB_BLOCK_DEFAULT 1024
#endif
constexpr int NIB_BINS_MAX = 262144;                   // largest range: 128 KB of LDS (the list's `nib` array is sized for it)
__global__ __launch_bounds__(1024) void k_nibble_hist(MinimizerList ml, uint32_t n_regions, MinimizerParams P,
                                                      uint32_t n_spectra, uint32_t n_parts, uint64_t n_reads, int nranges, int rlog) {
    const int32_t NIB_BINS = 1 << rlog, NIB_WORDS = NIB_BINS >> 3;       // bins per range / words of a range's part (launch parameter)
    const uint32_t bs = blockDim.x;
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
        for (; j + 3u * bs < n4; j += 4u * bs) {
            const uint4 q0 = k4[j], q1 = k4[j + bs], q2 = k4[j + 2u * bs], q3 = k4[j + 3u * bs];
            HULK_NIB1(q0.x) HULK_NIB1(q0.y) HULK_NIB1(q0.z) HULK_NIB1(q0.w)
            HULK_NIB1(q1.x) HULK_NIB1(q1.y) HULK_NIB1(q1.z) HULK_NIB1(q1.w)
            HULK_NIB1(q2.x) HULK_NIB1(q2.y) HULK_NIB1(q2.z) HULK_NIB1(q2.w)
            HULK_NIB1(q3.x) HULK_NIB1(q3.y) HULK_NIB1(q3.z) HULK_NIB1(q3.w)
        }
        for (; j < n4; j += bs) { const uint4 q0 = k4[j]; HULK_NIB1(q0.x) HULK_NIB1(q0.y) HULK_NIB1(q0.z) HULK_NIB1(q0.w) }
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
                                                      uint32_t n_spectra, uint32_t n_parts, int nranges, int rlog) {
    const int32_t NIB_BINS = 1 << rlog, NIB_WORDS = NIB_BINS >> 3;
    const int t = blockIdx.y;
    if (ml.nib_over[t]) return;
    const uint32_t slot = P.interval ? (uint32_t)(((uint64_t)t + P.ring_base) % P.ring_n) : P.ring_base;
    uint32_t *hist = hists + (size_t)slot * (size_t)P.num_bins;
    const int total_words = nranges * NIB_WORDS;
    const size_t pstride = (size_t)n_spectra * (size_t)nranges * NIB_WORDS;
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < total_words; w += gridDim.x * blockDim.x) {
        const int r = w / NIB_WORDS, wi = w - r * NIB_WORDS;
        const int32_t b0 = r * NIB_BINS + wi * 8;
        if (b0 >= P.num_bins) continue;
        uint32_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const uint32_t *src = ml.nib + ((size_t)t * (size_t)nranges + r) * NIB_WORDS + wi;
        // eight parts per step, all eight loads in flight (the kernel is nothing but these strided loads); byte-lane SWAR
        // sums of the even / odd nibbles: 8 x 15 = 120 fits a byte
        for (uint32_t p = 0; p < n_parts; p += 8) {
            uint32_t v[8];
#pragma unroll
            for (int x = 0; x < 8; x++) v[x] = p + x < n_parts ? src[(size_t)(p + x) * pstride] : 0u;
            uint32_t even = 0, odd = 0;
#pragma unroll
            for (int x = 0; x < 8; x++) { even += v[x] & 0x0F0F0F0Fu; odd += (v[x] >> 4) & 0x0F0F0F0Fu; }
#pragma unroll
            for (int q = 0; q < 4; q++) { c[2 * q] += (even >> (8 * q)) & 0xFFu; c[2 * q + 1] += (odd >> (8 * q)) & 0xFFu; }
        }
        if (b0 + 8 <= P.num_bins && (((uintptr_t)(hist + b0)) & 15u) == 0) {        // two 16-byte read-modify-writes
            uint4 *h4 = (uint4 *)(hist + b0);
            uint4 a0 = h4[0], a1 = h4[1];
            a0.x += c[0]; a0.y += c[1]; a0.z += c[2]; a0.w += c[3]; a1.x += c[4]; a1.y += c[5]; a1.z += c[6]; a1.w += c[7];
            h4[0] = a0; h4[1] = a1;
        } else {
#pragma unroll
            for (int q = 0; q < 8; q++) if (c[q] && b0 + q < P.num_bins) hist[b0 + q] += c[q];
        }
    }
}

// exact recount of the spectra whose 4-bit counters overflowed (flagged in nib_over by k_nibble_hist): every key of such a
// spectrum is added to it with a global atomic; without a flag (the normal case) every thread returns after one cached load
__global__ __launch_bounds__(256) void k_recount_flagged(MinimizerList ml, uint32_t n_regions, MinimizerParams P, uint32_t n_spectra,
                                                         uint64_t n_reads, uint32_t *__restrict__ hists) {
    uint32_t flagged = 0;
    for (uint32_t t = 0; t < n_spectra && t < 32u; t++) if (ml.nib_over[t]) flagged |= 1u << t;
    if (!flagged) return;
    const uint32_t *kl = ml.key;
    while (flagged) {
        const int t = __ffs((int)flagged) - 1;
        flagged &= flagged - 1u;
        uint64_t rd0 = 0, rd1 = n_reads;                              // reads of spectrum t, as in k_nibble_hist
        if (P.interval) {
            const uint64_t lo = (uint64_t)t * P.interval, hi = lo + P.interval;
            rd0 = lo > P.fill ? lo - P.fill : 0;
            rd1 = hi > P.fill ? hi - P.fill : 0;
            if (rd1 > n_reads) rd1 = n_reads;
        }
        if (rd0 >= rd1) continue;
        const uint32_t slot = P.interval ? (uint32_t)(((uint64_t)t + P.ring_base) % P.ring_n) : P.ring_base;
        const uint32_t g0 = (uint32_t)(rd0 / FAST_READS_PER_WAVE), g1 = (uint32_t)((rd1 - 1) / FAST_READS_PER_WAVE);
        const uint32_t a = ml.off[g0], b = ml.off[(g1 + 1 < n_regions ? g1 + 1 : n_regions)];
        uint32_t *h = hists + (size_t)slot * (size_t)P.num_bins;
        for (uint32_t i = a + blockIdx.x * blockDim.x + threadIdx.x; i < b; i += gridDim.x * blockDim.x) {
            const uint32_t k = kl[i];
            if ((k >> 20) == slot) atomicAdd(&h[k & 0xFFFFFu], 1u);
        }
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

__global__ void k_add_hist(uint32_t *hist, const uint32_t *add, int32_t n) {
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) hist[i] += add[i];
}

}  // namespace

// ---------------------------------------------------------------------------- host wrappers
// K1b: jump hash of the list (dense key array); K1c: spectrum ranges in LDS, merged without atomics
hipError_t launch_minimizer_post(hipStream_t s, uint64_t n_reads, MinimizerParams P, const MinimizerList &ml,
                                 uint32_t *d_hists, uint32_t *d_slow_list, uint32_t *d_slow_count, hipEvent_t jump_begin,
                                 hipEvent_t jump_end, hipEvent_t wait_before_spectra, hipEvent_t left_begin, hipEvent_t left_end) {
    if (n_reads == 0) return hipSuccess;
    hipError_t e = hipSuccess;
    const uint32_t n_regions = (uint32_t)((n_reads + FAST_READS_PER_WAVE - 1) / FAST_READS_PER_WAVE);
    const uint32_t nblk = (n_regions + 1023) / 1024;
    prof_mark(s, "k_region_bsum");
    hipLaunchKernelGGL(k_region_bsum, dim3(nblk), dim3(1024), 0, s, ml.cnt, ml.bsum, ml.dmask, ml.dsum, n_regions);
    prof_mark(s, "k_region_offsets");
    hipLaunchKernelGGL(k_region_offsets, dim3(nblk), dim3(1024), 0, s, ml.cnt, ml.bsum, ml.off, n_regions, ml.nib_over, ml.dmask,
                       ml.dsum, d_slow_list, d_slow_count);
    // k_jump_bin needs no LDS; a dummy allocation caps its occupancy so that the flush kernels of the
    // previous batch (other stream) find free wave slots next to it
    static int jump_lds = -1;
    if (jump_lds < 0) { const char *e = HULK_EXP_ENV("HULK_JUMP_LDS"); jump_lds = e ? atoi(e) : 0; }
    static int jump_c = -1;
    if (jump_c < 0) { const char *ec = HULK_EXP_ENV("HULK_JUMP_C"); jump_c = ec ? atoi(ec) : 0; }
    static int jump_cut = -1;
    if (jump_cut < 0) { const char *ec = HULK_EXP_ENV("HULK_JUMP_CUT"); jump_cut = ec ? atoi(ec) : 10; }
    const uint32_t cut = (jump_c || !ml.lo) ? 0u : (uint32_t)jump_cut;
    if (jump_begin) { e = hipEventRecord(jump_begin, s); if (e != hipSuccess) return e; }      // bench.py: k_jump_bin alone ...
    prof_mark(s, "k_jump_bin");
    hipLaunchKernelGGL(k_jump_bin, dim3((n_regions + 3) / 4), dim3(256), (size_t)jump_lds, s, ml, n_regions, P.num_bins, jump_c, cut);
    if (jump_end) { e = hipEventRecord(jump_end, s); if (e != hipSuccess) return e; }
    if (left_begin) { e = hipEventRecord(left_begin, s); if (e != hipSuccess) return e; }      // ... and k_jump_left alone
    if (cut) { prof_mark(s, "k_jump_left"); hipLaunchKernelGGL(k_jump_left, dim3((n_regions + 3) / 4), dim3(256), 0, s, ml, n_regions, P.num_bins); }
    if (left_end) { e = hipEventRecord(left_end, s); if (e != hipSuccess) return e; }
    // the kernels below write the spectra of the ring: the flush that last read them has to be done
    if (wait_before_spectra) { e = hipStreamWaitEvent(s, wait_before_spectra, 0); if (e != hipSuccess) return e; }
    const uint32_t n_spectra = P.interval ? (uint32_t)((P.fill + n_reads + P.interval - 1) / P.interval) : 1u;
    const int nranges = (P.num_bins + HIST_RANGE - 1) / HIST_RANGE;
    static int parts_target = -1;
    // workgroups per launch (swept 256..768: 110-123 us for histogram + merge, flat)
    if (parts_target < 0) { const char *ep = HULK_EXP_ENV("HULK_HIST_BLOCKS"); parts_target = ep ? atoi(ep) : 512; }
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
    if (use_nib < 0) use_nib = HULK_EXP_ENV("HULK_NO_NIBBLE") ? 0 : 1;
    if (use_nib && ml.nib && ml.nib_over) {
        // ~131 k keys per part (16 parts per 100k-read interval, swept 49k..197k): a 4-bit counter then overflows only on
        // grossly repetitive input, which the exact kernels below pick up
        // Range size and workgroup shape: a workgroup of 1024 threads that holds 2^18 four-bit counters (up to 128 KB of LDS)
        // reads every key once, but it only fits a CU that is free of k_minimizer_fast workgroups (4 x 39.7 KB) — and with two
        // work lanes the OTHER lane's are always there: rocprofv3 showed this kernel at 216 us per launch beside them against
        // 41 us alone, waiting for CUs (profiles/r04_kernel_stats.md).  Ranges of 2^16 bins (32 KB) in workgroups of 256 threads
        // slip in beside three of those; the keys are then read once per range (through the XCD's L2, see the kernel).
        static const int rlog = [] { const char *e = HULK_EXP_ENV("HULK_NIB_RLOG"); const int v = e ? atoi(e) : HULK_NIB_RLOG_DEFAULT; return v < 13 ? 13 : v > 18 ? 18 : v; }();
        static const int nib_block = [] { const char *e = HULK_EXP_ENV("HULK_NIB_BLOCK"); const int v = e ? atoi(e) : HULK_NIB_BLOCK_DEFAULT; return v == 256 || v == 512 ? v : 1024; }();
        const int32_t NIB_BINS = 1 << rlog, NIB_WORDS = NIB_BINS >> 3;
        const int nr = (P.num_bins + NIB_BINS - 1) / NIB_BINS;
        const int nr18 = (P.num_bins + NIB_BINS_MAX - 1) / NIB_BINS_MAX;
        const uint64_t rps = P.interval ? std::min<uint64_t>(P.interval, n_reads) : n_reads;
        // (per bin RANGE: with nr ranges a part's keys spread over nr workgroups, so a part is nr times as long — the mean
        //  count per 4-bit counter stays ~0.5, and k = 31 (4 ranges) builds 256 parts per batch instead of 1024: +3 %)
        static int keys_env = -1;
        if (keys_env < 0) { const char *ek = HULK_EXP_ENV("HULK_NIB_KEYS"); keys_env = ek ? atoi(ek) : 0; }
        const uint64_t keys_per_part = keys_env > 0 ? (uint64_t)keys_env : 131072ull * (uint64_t)nr18;   // (mean count per counter ~0.6 whatever the range size)
        uint32_t np = (uint32_t)((rps * 20 + keys_per_part - 1) / keys_per_part);
        if (np < 1) np = 1;
        // short intervals (a rank's slice of a strong-scaling run): still ~128 workgroups, one per CU would leave half the chip idle
        static int min_blocks = -1;
        if (min_blocks < 0) { const char *em = HULK_EXP_ENV("HULK_NIB_MIN_BLOCKS"); min_blocks = em ? atoi(em) : 128; }
        if (rps >= 4096 && (uint64_t)np * n_spectra * nr < (uint64_t)min_blocks) np = (uint32_t)(((uint64_t)min_blocks + (uint64_t)n_spectra * nr - 1) / ((uint64_t)n_spectra * nr));
        if (np > ml.nib_parts) np = ml.nib_parts;
        while (np > 1 && (uint64_t)np * n_spectra * nr > 8192) np--;
        const int words = ((std::min<int32_t>(P.num_bins, NIB_BINS) + 7) >> 3);
        static bool nib_attr = false;
        if (!nib_attr) {
            e = hipFuncSetAttribute((const void *)k_nibble_hist, hipFuncAttributeMaxDynamicSharedMemorySize, (NIB_BINS_MAX / 8) * 4);
            if (e != hipSuccess) return e;
            nib_attr = true;
        }
        const unsigned pg = (n_spectra * np + 7) / 8;
        prof_mark(s, "k_nibble_hist");
        hipLaunchKernelGGL(k_nibble_hist, dim3(8u * (unsigned)nr * pg), dim3(nib_block), (size_t)words * 4, s, ml, n_regions, P,
                           n_spectra, np, n_reads, nr, rlog);
        int nb = (nr * NIB_WORDS + 255) / 256; if (nb > 512) nb = 512;      // one word (8 bins) per thread up to 131072 words
        prof_mark(s, "k_nibble_merge");
        hipLaunchKernelGGL(k_nibble_merge, dim3(nb, n_spectra), dim3(256), 0, s, ml, d_hists, P, n_spectra, np, nr, rlog);
        // A spectrum in which a 4-bit counter overflowed (grossly repetitive input) is recounted exactly: k_nibble_merge left
        // it alone, k_recount_flagged adds its keys with global atomics.  Rare, so its shape is chosen for the common case,
        // in which it finds no flag: 256 light workgroups without LDS.  (Until round 4 the exact range histogram ran here with
        // an "only if flagged" test: 96 workgroups of 128 KB of LDS that return at once — but first have to be PLACED, and
        // beside the other lane's kernels that took 138 us of this lane's stream per batch.)
        prof_mark(s, "k_recount_flagged");
        hipLaunchKernelGGL(k_recount_flagged, dim3(256), dim3(256), 0, s, ml, n_regions, P, n_spectra, n_reads, d_hists);
        return hipGetLastError();
    }
    const unsigned pair_groups = (n_spectra * n_parts + 7) / 8;
    prof_mark(s, "k_range_hist");
    hipLaunchKernelGGL(k_range_hist, dim3(8u * (unsigned)nranges * pair_groups), dim3(1024), HIST_RANGE * 4, s, ml, n_regions,
                       ml.partial, P, n_spectra, n_parts, n_reads, nranges, nullptr, nullptr);
    int mb = (P.num_bins + 255) / 256; if (mb > 512) mb = 512;
    prof_mark(s, "k_merge_hist");
    hipLaunchKernelGGL(k_merge_hist, dim3(mb, n_spectra), dim3(256), 0, s, ml.partial, d_hists, P, n_spectra, n_parts, nullptr);
    return hipGetLastError();
}

hipError_t launch_add_hist(hipStream_t s, uint32_t *d_hist, const uint32_t *d_add, int32_t num_bins) {
    int blocks = (num_bins + 255) / 256; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_add_hist, dim3(blocks), dim3(256), 0, s, d_hist, d_add, num_bins);
    return hipGetLastError();
}

}  // namespace hulk
