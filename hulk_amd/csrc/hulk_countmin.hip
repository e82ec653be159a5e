// hulk_countmin.hip — KmerSpectrum.Cardinality and count-min Add() for a batch of spectra, bin order.
//   K2  k_count_used       the 1 % rule (kmerspectrum.go:53-55,84-96)
//   K3  k_cms_segsum/k_cms_base/k_cms_freq   src/countmin/countmin.go:103-147
//       k_elem_*/k_cmsd_*  the same with uniform scaling (0 < decay < 1)
#include "hulk_device.h"

#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <algorithm>
#include <mutex>

namespace hulk {
namespace {

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
//   k_cms_freq   : one workgroup per (segment, spectrum): waves 0..6 replay their row in bin order — one returning LDS
//                  atomic add per (row, bin): the LDS itself applies same-counter lanes in bin order — and meet in an LDS
//                  minimum per bin; wave 7 writes f / 1/f and wipes the spectrum.  No est arrays at all.
// ------------------------------------------------------------------------------------------
constexpr int CMS_SEGS = 16;          // bin segments per spectrum

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
                                                  const unsigned long long *__restrict__ base,
                                                  double *__restrict__ f64, float *__restrict__ rcp32,
                                                  int depth, int width, int seg_chunks, size_t row_stride,
                                                  DevState *st, FlushBatch fb) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int FG = 8, GB = FG * 64;                                          // chunks of 64 bins per barrier
    unsigned long long *lctr = (unsigned long long *)smem;                       // [depth][width]
    unsigned long long *smin = lctr + (size_t)depth * width;                     // [2][GB] minimum over the rows
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
        for (int i = tid; i < 2 * GB; i += blockDim.x) smin[i] = ~0ull;
    }
    __syncthreads();
    uint32_t *hist = hists + (size_t)slot * B;
    double *ft = f64 + (size_t)t * B;
    float *rt = rcp32 + (size_t)t * row_stride;
    const int64_t b0 = (int64_t)seg * seg_chunks * 64;
    const int ngroups = (seg_chunks + FG - 1) / FG;
    const uint16_t *pd = pos16 + (size_t)(d < depth ? d : 0) * B;
    unsigned long long *rc = lctr + (size_t)(d < depth ? d : 0) * width;
    // The LDS keeps the bin order by itself: ds_add_rtn_u64 returns the counter as it stood before this lane's add, same-
    // address lanes of one instruction are applied in ascending lane order, a wave's instructions in program order
    // (tools/ubench/lds_atomic_order.hip) — est = old + own count is the reference's counter after Add (countmin.go:122-127).
    // The rows meet in one LDS minimum per bin; the combiner wave is one group behind.  (Until round 5: a static "previous
    // lane on the same counter" table followed with register exchanges, per-row staging, 4 chunks per barrier: 117 us.)
    uint32_t nh[FG], np_[FG];
    auto fetch = [&](int g) {
#pragma unroll
        for (int c = 0; c < FG; c++) {
            const int ch = g * FG + c;
            const int64_t b = b0 + (int64_t)ch * 64 + lane;
            nh[c] = 0; np_[c] = 0;
            if (ch < seg_chunks && b < (int64_t)B) { nh[c] = hist[b]; if (d < depth) np_[c] = pd[b]; }
        }
    };
    fetch(0);
    uint32_t ph[FG];                                            // combiner: spectrum values of the group it finishes next
#pragma unroll
    for (int c = 0; c < FG; c++) ph[c] = 0;
    for (int g = 0; g <= ngroups; g++) {
        uint32_t hh[FG], pp[FG];
#pragma unroll
        for (int c = 0; c < FG; c++) { hh[c] = nh[c]; pp[c] = np_[c]; }
        if (g + 1 < ngroups) fetch(g + 1);
        if (d < depth && g < ngroups) {
            unsigned long long *my = smin + (size_t)(g & 1) * GB;
#pragma unroll
            for (int c = 0; c < FG; c++) {
                const uint32_t h = hh[c];
                if (h) atomicMin(&my[c * 64 + lane], atomicAdd(&rc[pp[c]], (unsigned long long)h) + h);
            }
        }
        if (d == depth && g > 0) {
            unsigned long long *src = smin + (size_t)((g - 1) & 1) * GB;
#pragma unroll
            for (int c = 0; c < FG; c++) {
                const int ch = (g - 1) * FG + c;
                const int64_t b = b0 + (int64_t)ch * 64 + lane;
                if (ch < seg_chunks && b < (int64_t)B) {
                    if (ph[c]) {
                        const unsigned long long mn = src[c * 64 + lane];
                        src[c * 64 + lane] = ~0ull;
                        const double f = (double)mn;
                        ft[b] = f; rt[b] = (float)(1.0 / f);
                        hist[b] = 0;                                 // Wipe (kmerspectrum.go:58-64)
                    } else { ft[b] = 0.0; rt[b] = __builtin_nanf(""); }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < FG; c++) ph[c] = hh[c];             // group g is finished by the combiner at g+1
        __syncthreads();
    }
}

// The fallback of k_cms_freq for a device whose LDS does NOT apply the same-address lanes of a returning atomic in ascending
// lane order (lds_order_verified below; HULK_FLAG_CMS_CHAIN forces it): the bin order inside a 64-bin chunk comes from the static
// table k_build_chains wrote — meta8 = {the nearest LOWER lane of the chunk on the same counter (64: none), bit 7: the last lane
// on its counter} — followed with ballots and register exchanges; the LDS is only read by the first lane of a counter's run
// and written by its last.  Same integers, same order of additions: bit-identical to k_cms_freq (tested), about twice its time.
__global__ __launch_bounds__(512) void k_cms_freq_chain(uint32_t *__restrict__ hists, const uint16_t *__restrict__ pos16,
                                                        const uint8_t *__restrict__ meta8,
                                                        const unsigned long long *__restrict__ base,
                                                        double *__restrict__ f64, float *__restrict__ rcp32,
                                                        int depth, int width, int seg_chunks, size_t row_stride,
                                                        DevState *st, FlushBatch fb) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int FG = 8, GB = FG * 64;
    unsigned long long *lctr = (unsigned long long *)smem;                       // [depth][width]
    unsigned long long *smin = lctr + (size_t)depth * width;                     // [2][GB] minimum over the rows
    const int seg = blockIdx.x, t = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, d = tid >> 6;
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
    if (st->skip_exact[fb.parity]) {
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
        for (int i = tid; i < 2 * GB; i += blockDim.x) smin[i] = ~0ull;
    }
    __syncthreads();
    uint32_t *hist = hists + (size_t)slot * B;
    double *ft = f64 + (size_t)t * B;
    float *rt = rcp32 + (size_t)t * row_stride;
    const int64_t b0 = (int64_t)seg * seg_chunks * 64;
    const int ngroups = (seg_chunks + FG - 1) / FG;
    const uint16_t *pd = pos16 + (size_t)(d < depth ? d : 0) * B;
    const uint8_t *md = meta8 + (size_t)(d < depth ? d : 0) * B;
    unsigned long long *rc = lctr + (size_t)(d < depth ? d : 0) * width;
    uint32_t ph[FG];
#pragma unroll
    for (int c = 0; c < FG; c++) ph[c] = 0;
    for (int g = 0; g <= ngroups; g++) {
        uint32_t hh[FG];
#pragma unroll
        for (int c = 0; c < FG; c++) {
            const int ch = g * FG + c;
            const int64_t b = b0 + (int64_t)ch * 64 + lane;
            const bool valid = g < ngroups && ch < seg_chunks && b < (int64_t)B;
            hh[c] = valid ? hist[b] : 0u;
            if (d < depth && g < ngroups && ch < seg_chunks) {                   // (wave-uniform)
                const uint32_t h = hh[c], p = valid ? pd[b] : 0u, m = valid ? md[b] : (64u | 0x80u);
                const uint32_t prev = m & 0x7fu;
                unsigned long long *my = smin + (size_t)(g & 1) * GB;
                bool ready = false; unsigned long long Sn = 0;
                if (prev >= 64u) {                                               // first lane of the chunk on this counter
                    Sn = rc[p] + h;
                    if (h) atomicMin(&my[c * 64 + lane], Sn);
                    ready = true;
                }
                unsigned long long done = __ballot(ready);
                while (done != ~0ull) {
                    const unsigned long long ps = __shfl(Sn, (int)(prev & 63u));
                    const bool pr = (done >> (prev & 63u)) & 1ull;
                    if (!ready && pr) {
                        Sn = ps + h;
                        if (h) atomicMin(&my[c * 64 + lane], Sn);
                        ready = true;
                    }
                    done = __ballot(ready);
                }
                if ((m & 0x80u) && valid) rc[p] = Sn;
            }
        }
        if (d == depth && g > 0) {
            unsigned long long *src = smin + (size_t)((g - 1) & 1) * GB;
#pragma unroll
            for (int c = 0; c < FG; c++) {
                const int ch = (g - 1) * FG + c;
                const int64_t b = b0 + (int64_t)ch * 64 + lane;
                if (ch < seg_chunks && b < (int64_t)B) {
                    if (ph[c]) {
                        const unsigned long long mn = src[c * 64 + lane];
                        src[c * 64 + lane] = ~0ull;
                        const double f = (double)mn;
                        ft[b] = f; rt[b] = (float)(1.0 / f);
                        hist[b] = 0;                                 // Wipe (kmerspectrum.go:58-64)
                    } else { ft[b] = 0.0; rt[b] = __builtin_nanf(""); }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < FG; c++) ph[c] = hh[c];
        __syncthreads();
    }
}

// Wave priority of the issue arbiter (0..3).  The count-min replay kernels with decay are chains of dependent LDS round trips, one
// workgroup per CU; beside them run kernels that fill the SIMDs' issue slots (k_minimizer_fast, k_jump_bin).  A replay wave that is
// ready should not queue behind eight of theirs.
__device__ __forceinline__ void set_wave_prio(int p) {
    if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else if (p >= 3) __builtin_amdgcn_s_setprio(3);
}
// ------------------------------------------------------------------------------------------
// K3 with uniform scaling (0 < decay < 1), bin-order form.  Counter (d,p) right after stream element j
// is C(j) = w*C(j-1) + (v_j if element j hits it).  Over a bin segment holding elements [e0,e1):
//     C(e1-1) = w^(e1-e0) * C(e0-1) + sum{ v_j * w^(e1-1-j) : hits }
//   k_cmsd_segsum : the sum (one w^x per element; wave d owns row d: ascending bin order, reproducible) and the factor per segment
//   k_cmsd_base   : C in front of every (spectrum, segment), advancing the persistent fp64 counters
//   k_cmsd_freq   : replay in bin order with LDS counters normalised to a moving base element; zero bins are
//                   transparent; same-counter lanes of a 64-bin chunk resolve in lane order
// (fp64 sums are re-associated w.r.t. the reference's step-by-step scaling: ~1e-13 relative)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_cmsd_segsum(const uint32_t *__restrict__ hists, const uint16_t *__restrict__ pos16,
                                                     const uint32_t *__restrict__ eidx, const uint32_t *__restrict__ etot,
                                                     double *__restrict__ segadd, double *__restrict__ segfac,
                                                     uint32_t *__restrict__ sege0, int depth, int width, int seg_chunks,
                                                     double omega, const DevState *st, FlushBatch fb) {
    extern __shared__ __align__(16) unsigned char smem[];
    double *ladd = (double *)smem;                                // [depth][width]
    set_wave_prio(fb.prio);
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
    // Deterministic sums (round 3; before, every thread added its bins' terms to all rows with fp64 LDS atomics, in an order
    // that depended on the timing of the 8 waves: weights reproducible to ~1e-13, not to the bit).  Now the terms of a group of
    // 512 bins are computed once, one per thread, into an LDS staging array, and row d's counters are touched by wave d ONLY:
    // it walks the group's 8 chunks in order and adds lane l's term to its counter with one LDS atomic per chunk — same-address
    // lanes of one instruction resolve in the hardware's fixed lane order, successive instructions of a wave in program order,
    // so every counter receives its terms in ascending bin order, the same in every run.  Zero bins are transparent.
    double *stage = ladd + (size_t)depth * width;                 // [2][512]
    const int d = tid >> 6, lane = tid & 63;
    const uint16_t *pd = pos16 + (size_t)(d < depth ? d : 0) * B;
    const int64_t bend = b1 < (int64_t)B ? b1 : (int64_t)B;
    const int ngroups = (int)((bend - b0 + 511) / 512);
    // software pipeline: the terms of group g+1 are computed (VALU: one exp per thread) behind the LDS atomics of group g
    // (issued first, fire and forget), one barrier per group
    auto term = [&](uint32_t h, uint32_t e) { return h ? (double)h * exp((double)(e1 - 1u - e) * lnw) : 0.0; };
    uint32_t nh = 0, ne = 0;
    { const int64_t b = b0 + tid; if (b < bend) { nh = hist[b]; ne = ei[b]; } }
    if (ngroups > 0) stage[tid] = term(nh, ne);
    { const int64_t b = b0 + 512 + tid; nh = 0; ne = 0; if (b < bend) { nh = hist[b]; ne = ei[b]; } }
    uint16_t pn[8];
    auto fetch_pos = [&](int g) {
#pragma unroll
        for (int c = 0; c < 8; c++) { const int64_t b = b0 + (int64_t)g * 512 + c * 64 + lane; pn[c] = (d < depth && b < bend) ? pd[b] : (uint16_t)0; }
    };
    fetch_pos(0);
    __syncthreads();
    for (int g = 0; g < ngroups; g++) {
        uint16_t pp[8];
#pragma unroll
        for (int c = 0; c < 8; c++) pp[c] = pn[c];
        if (g + 1 < ngroups) fetch_pos(g + 1);
        if (d < depth) {
            const double *sg = stage + (g & 1) * 512;
            double wv[8];
#pragma unroll
            for (int c = 0; c < 8; c++) wv[c] = sg[c * 64 + lane];
#pragma unroll
            for (int c = 0; c < 8; c++) if (wv[c] != 0.0) atomicAdd(&ladd[d * width + pp[c]], wv[c]);
        }
        if (g + 1 < ngroups) {
            const uint32_t h = nh, e = ne;
            { const int64_t b = b0 + (int64_t)(g + 2) * 512 + tid; nh = 0; ne = 0; if (b < bend) { nh = hist[b]; ne = ei[b]; } }
            stage[((g + 1) & 1) * 512 + tid] = term(h, e);
        }
        __syncthreads();
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

// Replay with decay.  Workgroup (segment, spectrum): wave d < depth replays row d of the count-min sketch over the
// segment's bins in order; the estimate of a bin is the minimum over the rows, which the row waves form with ONE 64-bit
// LDS atomic minimum per (row, bin) on a small staging array (the estimates are non-negative doubles, whose bit patterns
// order like unsigned integers; +inf = "bin not in the stream").  A barrier covers CMSD_FG = 8 chunks of 64 bins; the
// combiner wave (d == depth) reads one value per bin, writes f (fp64), 1/f (fp32), wipes the spectrum and resets the
// staging slot while the row waves are one group ahead.
//
// The row waves are a chain of dependent LDS round trips, one workgroup per CU (112 KB of counters), so the number of
// round trips per chunk IS the kernel's time.  A counter is kept ADDITIVELY normalised to a base element index,
// S = sum over its elements i of v_i * w^-(i - base), so that its value right after element j is S * w^(j - base): an element
// only ADDS g = v * w^-(j - base) to its counter.  Every CMSD_PERIOD elements the base moves on and the row's 2000
// counters are rescaled by w^PERIOD (the period keeps w^-(j - base) far below the fp64 range for any decay < 1).
//
// k_cmsd_freq (below) lets the LDS keep the bin order: ds_add_rtn_f64 returns the counter as it stood before this lane's add.
// k_cmsd_freq_chain (here) is its FALLBACK for a device whose LDS does not apply same-address lanes of one instruction in
// ascending lane order (lds_order_verified; HULK_FLAG_CMS_CHAIN forces it): the order inside a 64-bin chunk comes from
// k_build_chains' static table (meta8: nearest lower lane on the same counter, last-lane flag), followed with ballots and
// register exchanges; the same additions in the same order — bit-identical to k_cmsd_freq where the LDS does keep the order
// (tested) — at ~1.8x its time (517 -> 450 us was this form's round-3/4 history, docs/EXPERIMENTS.md).
constexpr int CMSD_FG = 8;            // chunks per barrier group of k_cmsd_freq
constexpr int CMSD_WAVE_PRIO = 0;     // s_setprio of the replay kernels (set_wave_prio)
__global__ __launch_bounds__(512) void k_cmsd_freq_chain(uint32_t *__restrict__ hists, const uint16_t *__restrict__ pos16,
                                                   const uint8_t *__restrict__ meta8, const uint32_t *__restrict__ eidx,
                                                   const uint32_t *__restrict__ sege0, const double *__restrict__ cstart,
                                                   double *__restrict__ f64, float *__restrict__ rcp32, int depth,
                                                   int width, int seg_chunks, size_t row_stride, double omega,
                                                   DevState *st, FlushBatch fb) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int GB = CMSD_FG * 64;
    constexpr unsigned long long INF_BITS = 0x7FF0000000000000ull;
    double *lval = (double *)smem;                                               // [depth][width] normalised counters
    unsigned long long *smin = (unsigned long long *)(lval + (size_t)depth * width);   // [2][GB] min over the rows, as bits
    __shared__ double tabf_lo[64], tabf_hi[66], tabi_lo[64], tabi_hi[66];          // w^x and w^-x for x = lo + 64 hi
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
    const double lnw = log(omega);                               // < 0
    int period = 4032;
    if (-lnw * (double)(period + 64) > 600.0) period = (int)(600.0 / -lnw) - 64;
    period &= ~63;
    if (period < 64) period = 64;
    {
        const double *bt = cstart + (((size_t)t * depth) * CMS_SEGS) * width;
        for (int i = tid; i < depth * width; i += blockDim.x) {
            const int dd = i / width, p = i - dd * width;
            lval[i] = bt[((size_t)dd * CMS_SEGS + seg) * width + p];
        }
        for (int i = tid; i < 2 * GB; i += blockDim.x) smin[i] = INF_BITS;
        if (tid < 64) { tabf_lo[tid] = exp((double)tid * lnw); tabi_lo[tid] = exp(-(double)tid * lnw); }
        if (tid >= 64 && tid < 64 + 66) { const int x = tid - 64; tabf_hi[x] = exp((double)(64 * x) * lnw); tabi_hi[x] = exp(-(double)(64 * x) * lnw); }
    }
    __syncthreads();
    uint32_t *hist = hists + (size_t)slot * B;
    const uint32_t *ei = eidx + (size_t)t * B;
    double *ft = f64 + (size_t)t * B;
    float *rt = rcp32 + (size_t)t * row_stride;
    const int64_t b0 = (int64_t)seg * seg_chunks * 64;
    long long base = (long long)sege0[(size_t)t * CMS_SEGS + seg] - 1;
    const double wperiod = exp((double)period * lnw);
    const int ngroups = (seg_chunks + CMSD_FG - 1) / CMSD_FG;
    const uint16_t *pd = pos16 + (size_t)(d < depth ? d : 0) * B;
    const uint8_t *md = meta8 + (size_t)(d < depth ? d : 0) * B;
    double *rv = lval + (size_t)(d < depth ? d : 0) * width;
    uint32_t nh[CMSD_FG], np_[CMSD_FG], nm[CMSD_FG], nj[CMSD_FG];
    auto fetch = [&](int g) {
#pragma unroll
        for (int c = 0; c < CMSD_FG; c++) {
            const int ch = g * CMSD_FG + c;
            const int64_t b = b0 + (int64_t)ch * 64 + lane;
            nh[c] = 0; np_[c] = 0; nm[c] = 64u | 0x80u; nj[c] = 0;
            if (d < depth && ch < seg_chunks && b < (int64_t)B) { nh[c] = hist[b]; np_[c] = pd[b]; nm[c] = md[b]; nj[c] = ei[b]; }
        }
    };
    fetch(0);
    uint32_t chh[CMSD_FG], cp[CMSD_FG], cm[CMSD_FG], cj[CMSD_FG];
    for (int g = 0; g <= ngroups; g++) {
#pragma unroll
        for (int c = 0; c < CMSD_FG; c++) { chh[c] = nh[c]; cp[c] = np_[c]; cm[c] = nm[c]; cj[c] = nj[c]; }
        if (g + 1 < ngroups) fetch(g + 1);
        if (d < depth && g < ngroups) {
            unsigned long long *my = smin + (size_t)(g & 1) * GB;
#pragma unroll
            for (int c = 0; c < CMSD_FG; c++) {
                const int ch = g * CMSD_FG + c;
                if (ch >= seg_chunks) break;
                const int64_t b = b0 + (int64_t)ch * 64 + lane;
                const uint32_t h = chh[c], p = cp[c], m = cm[c]; const long long j = (long long)cj[c];
                {
                    const long long jfirst = (long long)__builtin_amdgcn_readfirstlane((int)cj[c]);
                    while (jfirst - base > (long long)period) {
                        for (int i = lane; i < width; i += 64) rv[i] *= wperiod;
                        base += period;
                    }
                }
                const uint32_t x = h ? (uint32_t)(j - base) : 0u;                // 1 .. period + 64 (unused for bins not in the stream)
                const double wi = tabi_lo[x & 63u] * tabi_hi[x >> 6], wf = tabf_lo[x & 63u] * tabf_hi[x >> 6];
                const double gv = (double)h * wi;
                // resolve the lanes in same-counter order: a lane is computed once its predecessor is (whether it is comes
                // from the wave's ballot of finished lanes).  `before` is what ds_add_rtn_f64 returns in k_cmsd_freq.
                const uint32_t prev = m & 0x7fu;
                bool ready = false; double Sn = 0.0;
                if (prev >= 64u) {                                  // first lane of the chunk on this counter: LDS state
                    const double before = rv[p];
                    if (h) { Sn = before + gv; atomicMin(&my[c * 64 + lane], (unsigned long long)__double_as_longlong(Sn * wf)); }
                    else Sn = before;
                    ready = true;
                }
                unsigned long long done = __ballot(ready);
                while (done != ~0ull) {
                    const double before = __shfl(Sn, (int)(prev & 63u));
                    const bool pr = (done >> (prev & 63u)) & 1ull;
                    if (!ready && pr) {
                        if (h) { Sn = before + gv; atomicMin(&my[c * 64 + lane], (unsigned long long)__double_as_longlong(Sn * wf)); }
                        else Sn = before;
                        ready = true;
                    }
                    done = __ballot(ready);
                }
                if ((m & 0x80u) && b < (int64_t)B) rv[p] = Sn;
            }
        }
        if (d == depth && g > 0) {
            unsigned long long *src = smin + (size_t)((g - 1) & 1) * GB;
#pragma unroll
            for (int c = 0; c < CMSD_FG; c++) {
                const int ch = (g - 1) * CMSD_FG + c;
                const int64_t b = b0 + (int64_t)ch * 64 + lane;
                if (ch < seg_chunks && b < (int64_t)B) {
                    const unsigned long long bits = src[c * 64 + lane];
                    src[c * 64 + lane] = INF_BITS;
                    if (bits != INF_BITS) {
                        const double mn = __longlong_as_double((long long)bits);
                        ft[b] = mn; rt[b] = (float)(1.0 / mn);
                    } else { ft[b] = 0.0; rt[b] = __builtin_nanf(""); }
                    hist[b] = 0;
                }
            }
        }
        __syncthreads();
    }
}


// The same replay with the counter kept ADDITIVELY normalised, S = sum over its elements i of v_i * w^-(i - base), so that its
// value right after element j is S * w^(j - base): an element then only ADDS g = v * w^-(j - base) to its counter, and the
// LDS does the bin-order bookkeeping by itself — ds_add_rtn_f64 returns the counter as it stood before this lane's add,
// same-address lanes of one instruction are applied in ascending lane order and the instructions of a wave in program order
// (tools/ubench/lds_atomic_order.hip checks both on the chip).  No "previous lane on the same counter" table, no ballots,
// no register exchanges, no read-then-write hazard between consecutive chunks: the eight atomics of a group are in flight
// together.  ~15 VALU + 6 LDS instructions per (row, chunk) where the chain form has ~90 VALU: 450 -> see
// profiles/r05_c3_kernel_stats_serial.md.  (Rounding: C = (S + v*wi) * wf against the chain form's S*wf + v — both are
// re-associations of the reference's step-by-step scaling, ~1e-13 relative; bit-reproducible from run to run.)
template <int FG>
__global__ __launch_bounds__(512) void k_cmsd_freq(uint32_t *__restrict__ hists, const uint16_t *__restrict__ pos16,
                                                   const uint32_t *__restrict__ eidx, const uint32_t *__restrict__ sege0,
                                                   const double *__restrict__ cstart, double *__restrict__ f64,
                                                   float *__restrict__ rcp32, int depth, int width, int seg_chunks,
                                                   size_t row_stride, double omega, DevState *st, FlushBatch fb) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int GB = FG * 64;
    constexpr unsigned long long INF_BITS = 0x7FF0000000000000ull;
    double *lval = (double *)smem;                                               // [depth][width] normalised counters
    unsigned long long *smin = (unsigned long long *)(lval + (size_t)depth * width);   // [2][GB] min over the rows, as bits
    __shared__ double tabf_lo[64], tabf_hi[66], tabi_lo[64], tabi_hi[66];          // w^x and w^-x for x = lo + 64 hi
    set_wave_prio(fb.prio);
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
    const double lnw = log(omega);                               // < 0
    // elements per base: |ln w| * (period + 64) <= 600  =>  w^-(j - base) <= e^600 (counters stay below ~1e270)
    int period = 4032;
    if (-lnw * (double)(period + 64) > 600.0) period = (int)(600.0 / -lnw) - 64;
    period &= ~63;
    if (period < 64) period = 64;
    {
        const double *bt = cstart + (((size_t)t * depth) * CMS_SEGS) * width;
        for (int i = tid; i < depth * width; i += blockDim.x) {
            const int dd = i / width, p = i - dd * width;
            lval[i] = bt[((size_t)dd * CMS_SEGS + seg) * width + p];       // value as of element e0 - 1: see `base` below
        }
        for (int i = tid; i < 2 * GB; i += blockDim.x) smin[i] = INF_BITS;
        if (tid < 64) { tabf_lo[tid] = exp((double)tid * lnw); tabi_lo[tid] = exp(-(double)tid * lnw); }
        if (tid >= 64 && tid < 64 + 66) { const int x = tid - 64; tabf_hi[x] = exp((double)(64 * x) * lnw); tabi_hi[x] = exp(-(double)(64 * x) * lnw); }
    }
    __syncthreads();
    uint32_t *hist = hists + (size_t)slot * B;
    const uint32_t *ei = eidx + (size_t)t * B;
    double *ft = f64 + (size_t)t * B;
    float *rt = rcp32 + (size_t)t * row_stride;
    const int64_t b0 = (int64_t)seg * seg_chunks * 64;
    // a counter holding S stands for the value S * w^(j - base) right after element j has been added (cstart = the value right
    // after element e0 - 1: S = cstart at base = e0 - 1)
    long long base = (long long)sege0[(size_t)t * CMS_SEGS + seg] - 1;
    const double wperiod = exp((double)period * lnw);
    const int ngroups = (seg_chunks + FG - 1) / FG;
    const uint16_t *pd = pos16 + (size_t)(d < depth ? d : 0) * B;
    double *rv = lval + (size_t)(d < depth ? d : 0) * width;
    // row waves: the three per-bin inputs of the WHOLE next group (8 chunks = 24 loads per lane) are requested before the
    // current group is computed: with one workgroup per CU nothing else hides their latency
    uint32_t nh[FG], np_[FG], nj[FG];
    auto fetch = [&](int g) {
#pragma unroll
        for (int c = 0; c < FG; c++) {
            const int ch = g * FG + c;
            const int64_t b = b0 + (int64_t)ch * 64 + lane;
            nh[c] = 0; np_[c] = 0; nj[c] = 0;
            if (d < depth && ch < seg_chunks && b < (int64_t)B) { nh[c] = hist[b]; np_[c] = pd[b]; nj[c] = ei[b]; }
        }
    };
    fetch(0);
    uint32_t chh[FG], cp[FG], cj[FG];
    for (int g = 0; g <= ngroups; g++) {
#pragma unroll
        for (int c = 0; c < FG; c++) { chh[c] = nh[c]; cp[c] = np_[c]; cj[c] = nj[c]; }
        if (g + 1 < ngroups) fetch(g + 1);
        if (d < depth && g < ngroups) {
            unsigned long long *my = smin + (size_t)(g & 1) * GB;
#pragma unroll
            for (int c = 0; c < FG; c++) {
                const int ch = g * FG + c;
                if (ch >= seg_chunks) continue;
                const uint32_t h = chh[c], p = cp[c]; const long long j = (long long)cj[c];
                // move the base on when the chunk's elements would leave the tables (wave-uniform: element indices
                // ascend with the lane; lane 0 holds the chunk's first)
                {
                    const long long jfirst = (long long)__builtin_amdgcn_readfirstlane((int)cj[c]);
                    while (jfirst - base > (long long)period) {
                        for (int i = lane; i < width; i += 64) rv[i] *= wperiod;
                        base += period;
                    }
                }
                if (h) {
                    const uint32_t x = (uint32_t)(j - base);                     // 1 .. period + 64
                    const double wi = tabi_lo[x & 63u] * tabi_hi[x >> 6], wf = tabf_lo[x & 63u] * tabf_hi[x >> 6];
                    const double gv = (double)h * wi;
                    const double before = atomicAdd(&rv[p], gv);                 // ds_add_rtn_f64: the counter in bin order
                    const double C = (before + gv) * wf;
                    atomicMin(&my[c * 64 + lane], (unsigned long long)__double_as_longlong(C));
                }
            }
        }
        if (d == depth && g > 0) {
            unsigned long long *src = smin + (size_t)((g - 1) & 1) * GB;
#pragma unroll
            for (int c = 0; c < FG; c++) {
                const int ch = (g - 1) * FG + c;
                const int64_t b = b0 + (int64_t)ch * 64 + lane;
                if (ch < seg_chunks && b < (int64_t)B) {
                    const unsigned long long bits = src[c * 64 + lane];
                    src[c * 64 + lane] = INF_BITS;
                    if (bits != INF_BITS) {
                        const double mn = __longlong_as_double((long long)bits);
                        ft[b] = mn; rt[b] = (float)(1.0 / mn);
                    } else { ft[b] = 0.0; rt[b] = __builtin_nanf(""); }
                    hist[b] = 0;
                }
            }
        }
        __syncthreads();
    }
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
                                                    int nblk, FlushBatch fb, DevState *st) {
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
    if (blk == 0 && tid == 0) {
        const unsigned total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        etot[t] = total;
        // the number of elements IS KmerSpectrum.Cardinality() (kmerspectrum.go:53-55): with decay the flush takes the 1 %
        // rule's count from here instead of a k_count_used pass of its own (26 us per 16 spectra at k = 31)
        st->used[fb.parity][ring_slot(fb, t)] = total;
    }
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

}  // namespace

// ---------------------------------------------------------------------------- host wrappers
hipError_t launch_count_used(hipStream_t s, const uint32_t *d_hists, DevState *st, const FlushBatch &fb) {
    int blocks = (fb.num_bins + 2047) / 2048; if (blocks > 128) blocks = 128;
    prof_mark(s, "k_count_used");
    hipLaunchKernelGGL(k_count_used, dim3(blocks, fb.count), dim3(256), 0, s, d_hists, st, fb);
    return hipGetLastError();
}

hipError_t launch_cms_binorder(hipStream_t s, uint32_t *d_hists, const uint16_t *d_pos16, const uint8_t *d_meta8,
                               unsigned long long *d_ctr, uint32_t *d_segsum, unsigned long long *d_base,
                               double *d_f64, float *d_rcp32, int depth, int width, size_t row_stride,
                               DevState *st, const FlushBatch &fb, bool chain) {
    const int chunks = (fb.num_bins + 63) / 64;
    const int seg_chunks = (chunks + CMS_SEGS - 1) / CMS_SEGS;
    const size_t lds1 = (size_t)depth * width * 4;
    const size_t lds3 = (size_t)depth * width * 8 + (size_t)2 * 8 * 64 * 8;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)k_cms_freq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_cms_freq_chain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    prof_mark(s, "k_cms_segsum");
    hipLaunchKernelGGL(k_cms_segsum, dim3(CMS_SEGS, fb.count), dim3(512), lds1, s, d_hists, d_pos16, d_segsum, depth, width,
                       seg_chunks, st, fb);
    prof_mark(s, "k_cms_base");
    hipLaunchKernelGGL(k_cms_base, dim3((depth * width + 255) / 256), dim3(256), 0, s, d_segsum, d_ctr, d_base, depth, width, st, fb);
    if (chain) {
        prof_mark(s, "k_cms_freq_chain");
        hipLaunchKernelGGL(k_cms_freq_chain, dim3(CMS_SEGS, fb.count), dim3(512), lds3, s, d_hists, d_pos16, d_meta8, d_base, d_f64,
                           d_rcp32, depth, width, seg_chunks, row_stride, st, fb);
    }
    else {
        prof_mark(s, "k_cms_freq");
        hipLaunchKernelGGL(k_cms_freq, dim3(CMS_SEGS, fb.count), dim3(512), lds3, s, d_hists, d_pos16, d_base, d_f64,
                           d_rcp32, depth, width, seg_chunks, row_stride, st, fb);
    }
    return hipGetLastError();
}

size_t cms_binorder_entries(int depth, int width) { return (size_t)depth * CMS_SEGS * width; }

hipError_t launch_cmsd_binorder(hipStream_t s, uint32_t *d_hists, const uint16_t *d_pos16, const uint8_t *d_meta8,
                                const uint32_t *d_eidx, const uint32_t *d_etot, double *d_ctrd, double *d_segadd,
                                double *d_segfac, uint32_t *d_sege0, double *d_cstart, double *d_f64, float *d_rcp32,
                                int depth, int width, size_t row_stride, double omega, DevState *st, const FlushBatch &fb_in,
                                hipEvent_t freq_begin, hipEvent_t freq_end, bool chain_form) {
    FlushBatch fb = fb_in;
    const int chunks = (fb.num_bins + 63) / 64;
    const int seg_chunks = (chunks + CMS_SEGS - 1) / CMS_SEGS;
    if (depth > 8) return hipErrorInvalidValue;                     // k_cmsd_segsum: one wave per row, 8 waves
    { static const char *e = HULK_EXP_ENV("HULK_CMSD_PRIO"); fb.prio = e ? atoi(e) : CMSD_WAVE_PRIO; }
    const size_t lds1 = (size_t)depth * width * 8 + (size_t)2 * 512 * 8;
    const size_t lds3 = (size_t)depth * width * 8 + (size_t)2 * CMSD_FG * 64 * 8;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)k_cmsd_segsum, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_cmsd_freq<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_cmsd_freq<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds3 + (size_t)2 * 8 * 64 * 8));
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_cmsd_freq_chain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    prof_mark(s, "k_cmsd_segsum");
    hipLaunchKernelGGL(k_cmsd_segsum, dim3(CMS_SEGS, fb.count), dim3(512), lds1, s, d_hists, d_pos16, d_eidx, d_etot, d_segadd,
                       d_segfac, d_sege0, depth, width, seg_chunks, omega, st, fb);
    prof_mark(s, "k_cmsd_base");
    hipLaunchKernelGGL(k_cmsd_base, dim3((depth * width + 255) / 256), dim3(256), 0, s, d_segadd, d_segfac, d_ctrd, d_cstart,
                       depth, width, st, fb);
    if (freq_begin) { const hipError_t e = hipEventRecord(freq_begin, s); if (e != hipSuccess) return e; }   // bench.py: k_cmsd_freq alone
    const bool chain = chain_form || HULK_EXP_ENV("HULK_CMSD_CHAIN") != nullptr;   // the fallback (lds_order_verified / HULK_FLAG_CMS_CHAIN)
    static const bool fg16 = HULK_EXP_ENV("HULK_CMSD_FG16") != nullptr;        // 16 chunks per barrier group (A/B)
    if (chain) {
        prof_mark(s, "k_cmsd_freq_chain");
        hipLaunchKernelGGL(k_cmsd_freq_chain, dim3(CMS_SEGS, fb.count), dim3(512), lds3, s, d_hists, d_pos16, d_meta8, d_eidx, d_sege0,
                           d_cstart, d_f64, d_rcp32, depth, width, seg_chunks, row_stride, omega, st, fb);
    }
    else if (fg16) {
        prof_mark(s, "k_cmsd_freq");
        hipLaunchKernelGGL(k_cmsd_freq<16>, dim3(CMS_SEGS, fb.count), dim3(512), lds3 + (size_t)2 * 8 * 64 * 8, s, d_hists, d_pos16, d_eidx, d_sege0,
                           d_cstart, d_f64, d_rcp32, depth, width, seg_chunks, row_stride, omega, st, fb);
    }
    else {
        prof_mark(s, "k_cmsd_freq");
        hipLaunchKernelGGL(k_cmsd_freq<8>, dim3(CMS_SEGS, fb.count), dim3(512), lds3, s, d_hists, d_pos16, d_eidx, d_sege0,
                           d_cstart, d_f64, d_rcp32, depth, width, seg_chunks, row_stride, omega, st, fb);
    }
    if (freq_end) { const hipError_t e = hipEventRecord(freq_end, s); if (e != hipSuccess) return e; }
    return hipGetLastError();
}

hipError_t launch_elem_index(hipStream_t s, const uint32_t *d_hists, uint32_t *d_blkcnt, uint32_t *d_eidx,
                             uint32_t *d_etot, const FlushBatch &fb, DevState *st) {
    const int nblk = (fb.num_bins + EIDX_BLOCK - 1) / EIDX_BLOCK;
    prof_mark(s, "k_elem_count");
    hipLaunchKernelGGL(k_elem_count, dim3(nblk, fb.count), dim3(256), 0, s, d_hists, d_blkcnt, nblk, fb);
    prof_mark(s, "k_elem_index");
    hipLaunchKernelGGL(k_elem_index, dim3(nblk, fb.count), dim3(256), 0, s, d_hists, d_blkcnt, d_eidx, d_etot, nblk, fb, st);
    return hipGetLastError();
}

int elem_index_blocks(int32_t num_bins) { return (num_bins + EIDX_BLOCK - 1) / EIDX_BLOCK; }

// ------------------------------------------------------------------------------------------
// The hardware property k_cms_freq / k_cmsd_freq / k_cmsd_segsum rest on, checked on the device itself once per process and
// device (hulk_create): for ONE returning LDS atomic add (ds_add_rtn_u64 / ds_add_rtn_f64) whose lanes hit the same address, the
// value a lane gets back is the value before the instruction plus the operands of the LOWER lanes on that address (ascending
// lane order), and the instructions of a wave are applied in program order.  Patterns: all 64 lanes on one address, pairs,
// pseudo-random partitions into 1..61 groups; two instructions back to back on different partitions.  Every lane checks its own
// returned value against the sum it can form from the wave's operands (readlane).  out[0] / out[1]: lanes out of order for
// u64 / f64.  (tools/ubench/lds_atomic_order.hip is the stand-alone form of the same test.)
// ------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(64) void k_lds_order_probe(unsigned int *out, int rounds, unsigned int sabotage) {
    __shared__ unsigned long long su[64];
    __shared__ double sd[64];
    const int l = threadIdx.x;
    unsigned bad_u = 0, bad_d = 0;
    uint32_t rng = 0x9E3779B9u * (uint32_t)(l + 1);
    for (int r = 0; r < rounds; r++) {
        const int mode = r & 3;
        rng = rng * 1664525u + 1013904223u; const uint32_t ra = rng >> 8;
        rng = rng * 1664525u + 1013904223u; const uint32_t rb = rng >> 8;
        // (range reduction by multiply-shift, not `ra % m`: for operands it can prove to be below 2^24 hipcc 7.2 expands the
        // remainder through v_rcp_iflag_f32 and never corrects an OVER-estimated quotient — x = q * m - 1 near 2^24 comes back as
        // 0xFFFFFF; the first version of this probe indexed the LDS with that and "found" 32 lanes out of order.
        // tools/ubench/urem24_check.hip, profiles/r06_urem24.txt.  No kernel of the library has such an operand pair.)
        const int p0 = mode == 0 ? 0 : mode == 1 ? l / 2 : (int)((ra * (uint32_t)(1 + r % 61)) >> 24);
        const int p1 = mode == 0 ? 0 : mode == 1 ? (63 - l) / 2 : (int)((rb * (uint32_t)(1 + r % 59)) >> 24);
        const unsigned long long v0 = 1ull + (unsigned)l, v1 = 1000ull + (unsigned)l;
        su[l] = 7ull * (unsigned)l; sd[l] = (double)(7 * l);
        __syncthreads();
        const unsigned long long a_u = atomicAdd(&su[p0], v0);
        const unsigned long long b_u = atomicAdd(&su[p1], v1);
        const double a_d = atomicAdd(&sd[p0], (double)v0);
        const double b_d = atomicAdd(&sd[p1], (double)v1);
        // expectation: start value + lower lanes of the same instruction (+ every lane of the earlier instruction) on that address
        unsigned long long ea = 7ull * (unsigned)p0, eb = 7ull * (unsigned)p1;
        for (int j = 0; j < 64; j++) {
            const int q0 = __builtin_amdgcn_readlane(p0, j), q1 = __builtin_amdgcn_readlane(p1, j);
            if (q0 == p0 && j < l) ea += 1ull + (unsigned)j;
            if (q0 == p1) eb += 1ull + (unsigned)j;
            if (q1 == p1 && j < l) eb += 1000ull + (unsigned)j;
        }
        if (sabotage) ea += (unsigned)(l == 5);                   // (self-test of the checker: HULK_LDS_PROBE_SABOTAGE, profiling build)
        bad_u += (a_u != ea) + (b_u != eb);
        bad_d += (a_d != (double)ea) + (b_d != (double)eb);       // (sums of small integers: exact in fp64)
        __syncthreads();
    }
    for (int off = 32; off; off >>= 1) { bad_u += __shfl_xor(bad_u, off); bad_d += __shfl_xor(bad_d, off); }
    if (l == 0) { out[0] = bad_u; out[1] = bad_d; }
}
}  // namespace

// 1: the LDS applies same-address lanes in ascending lane order (what every gfx950 measured so far does), 0: it does not — the
// count-min replay then runs k_cms_freq_chain / k_cmsd_freq_chain —, < 0: the probe could not run (hipError_t negated).
int lds_order_verified(int device) {
    static std::mutex mu;
    static int cache[64];
    static bool init = false;
    std::lock_guard<std::mutex> lk(mu);
    if (!init) { for (int &x : cache) x = -1000000; init = true; }
    if (device >= 0 && device < 64 && cache[device] != -1000000) return cache[device];
    unsigned int *d_out = nullptr, h[2] = {1, 1};
    hipError_t e = hipMalloc((void **)&d_out, 8);
    if (e == hipSuccess) {
        const unsigned sabotage = HULK_EXP_ENV("HULK_LDS_PROBE_SABOTAGE") != nullptr;
        hipLaunchKernelGGL(k_lds_order_probe, dim3(1), dim3(64), 0, 0, d_out, 488, sabotage);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(h, d_out, 8, hipMemcpyDeviceToHost);
        (void)hipFree(d_out);
    }
    const int res = e != hipSuccess ? -(int)e : (h[0] == 0 && h[1] == 0) ? 1 : 0;
    if (res == 0)
        fprintf(stderr, "libhulkhip: device %d: the LDS does not apply same-address lanes of a returning atomic in ascending lane order "
                        "(%u u64, %u f64 lanes out of order): the count-min replay uses the chain-form kernels\n", device, h[0], h[1]);
    if (device >= 0 && device < 64 && res >= 0) cache[device] = res;
    return res;
}

// Static chain tables of the bin-order count-min kernels (built once per context): for row d the counter position
// g = jump(bin + d*bin, width) of every bin (countmin.go:122-125), and per 64-bin chunk which earlier lane of the chunk
// hits the same counter (bits 0-6, 64 = none) and whether the bin is the last one of the chunk on its counter (bit 7).
// One wave per (row, chunk).  (A host loop did this until round 2: 7 * k^4 jump hashes, 70 ms of every hulk_create at k = 21.)
namespace {
__global__ __launch_bounds__(256) void k_build_chains(uint16_t *__restrict__ pos16, uint8_t *__restrict__ meta8, int32_t B, int width) {
    const int lane = lane_id();
    const int32_t chunk = (int32_t)blockIdx.x * 4 + (int32_t)(threadIdx.x >> 6);
    const int d = (int)blockIdx.y;
    const int32_t b = chunk * 64 + lane;
    if (chunk * 64 >= B) return;
    const bool valid = b < B;
    const uint32_t pos = valid ? (uint32_t)jump_hash((uint64_t)b + (uint64_t)d * (uint64_t)b, width) : 0xffffffffu;
    int prev = 64; bool later = false;
#pragma unroll
    for (int j = 0; j < 64; j++) {
        const uint32_t pj = (uint32_t)__builtin_amdgcn_readlane((int)pos, j);
        if (pj == pos && j < lane) prev = j;
        if (pj == pos && j > lane) later = true;
    }
    if (valid) {
        pos16[(size_t)d * (size_t)B + (size_t)b] = (uint16_t)pos;
        meta8[(size_t)d * (size_t)B + (size_t)b] = (uint8_t)prev | (later ? 0u : 0x80u);
    }
}
}  // namespace

hipError_t launch_build_chains(hipStream_t s, uint16_t *d_pos16, uint8_t *d_meta8, int32_t num_bins, int depth, int width) {
    const int chunks = (num_bins + 63) / 64;
    hipLaunchKernelGGL(k_build_chains, dim3((chunks + 3) / 4, depth), dim3(256), 0, s, d_pos16, d_meta8, num_bins, width);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// hulk_step_sharded, delta exchange.  In the steady state of a stream (no element of the step can lower any weight on
// any rank: k_flush_decide's whole-batch bound, evaluated one step earlier) all a flush still changes is the count-min
// counters (countmin.go:122-127), the element count and the 1 % rule (kmerspectrum.go:84-96) — then it wipes the spectra
// (kmerspectrum.go:58-64).  A rank therefore reduces each of ITS intervals to the increments it causes:
//   k_shard_local : workgroup (bin segment, interval): waves 0..6 add the segment's counts to their row's 2000 counters
//                   in LDS (as k_cms_segsum), wave 7 counts the used bins; the workgroup adds its sums to the interval's
//                   delta vector (consecutive addresses: coalesced atomics) and wipes the segment.  One pass, one launch.
//   k_shard_apply : after the all-gather of every rank's {header, deltas}: thread per counter walks the intervals of the
//                   step in stream order — rank 0's, rank 1's, ... — applies flush_go()'s rule to each and adds.
// Unsigned integer sums: the counters equal those of the spectra exchange bit for bit.
// ------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(512) void k_shard_local(uint32_t *__restrict__ hists, const uint16_t *__restrict__ pos16,
                                                     uint32_t *__restrict__ hdr, uint32_t *__restrict__ delta, int depth,
                                                     int width, int seg_chunks, FlushBatch fb) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *lctr = (uint32_t *)smem;                           // [depth][width]
    const int seg = blockIdx.x, t = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, d = tid >> 6;  // waves 0..depth-1: rows; wave depth: used bins
    for (int i = tid; i < depth * width; i += blockDim.x) lctr[i] = 0;
    __syncthreads();
    const size_t B = (size_t)fb.num_bins;
    uint32_t *hist = hists + (size_t)ring_slot(fb, t) * B;
    const int64_t b0 = (int64_t)seg * seg_chunks * 64;
    if (d < depth) {
        const uint16_t *pd = pos16 + (size_t)d * B;
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
    } else if (d == depth) {
        unsigned cnt = 0;
        for (int c0 = 0; c0 < seg_chunks; c0++) {
            const int64_t b = b0 + (int64_t)c0 * 64 + lane;
            if (b < (int64_t)B) cnt += hist[b] != 0;
        }
        for (int off = 32; off; off >>= 1) cnt += __shfl_xor(cnt, off);
        if (lane == 0 && cnt) atomicAdd(&hdr[SHARD_USED + t], cnt);
    }
    __syncthreads();
    uint32_t *out = delta + (size_t)t * depth * width;
    for (int i = tid; i < depth * width; i += blockDim.x) { const uint32_t v = lctr[i]; if (v) atomicAdd(&out[i], v); }
    const int64_t w1 = b0 + (int64_t)seg_chunks * 64;
    for (int64_t b = b0 + tid; b < w1 && b < (int64_t)B; b += blockDim.x) hist[b] = 0;      // Wipe
}

__global__ __launch_bounds__(256) void k_shard_apply(const uint32_t *__restrict__ hdr_all, const uint32_t *__restrict__ delta_all,
                                                     unsigned long long *__restrict__ ctr, int ncounters, uint32_t world,
                                                     uint32_t T, uint32_t step_intervals, int32_t num_bins, DevState *st,
                                                     uint32_t step_tag) {
    // grid = (counters / 256, ranks): a workgroup adds ONE rank's intervals to its 256 counters (integer sums: the order of
    // the ranks' atomic adds does not matter); the rank's header words come in with one load per lane
    __shared__ uint32_t h[SHARD_HDR];
    const uint32_t r = blockIdx.y;
    if (threadIdx.x < SHARD_HDR) h[threadIdx.x] = hdr_all[(size_t)r * SHARD_HDR + threadIdx.x];
    __syncthreads();
    const uint32_t lo = r * T;                                                 // rank r holds intervals [r*T, r*T + cnt) of the step
    const uint32_t cnt = step_intervals <= lo ? 0u : (step_intervals - lo < T ? step_intervals - lo : T);
    // flush_go()'s rule per interval
    uint32_t gomask = 0; unsigned long long elems = 0; bool few = false;
    for (uint32_t t = 0; t < cnt; t++) {
        const uint32_t used = h[SHARD_USED + t];
        if (used == 0) continue;                                               // boss.go:118: nothing to flush
        if ((double)used / (double)num_bins < 0.01) { few = true; continue; }  // kmerspectrum.go:88-96 ("not used yet")
        gomask |= 1u << t;
        elems += used;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        if (h[SHARD_TAG] != step_tag) set_error(st, -37);                      // HULK_ERR_COMM: a block that is not of this step (the seal: hulk_comm.hip)
        if (few) set_error(st, -5);
        if (elems) atomicAdd(&st->n_elements, elems);
    }
    if (i >= ncounters || !gomask) return;
    const uint32_t *dr = delta_all + (size_t)r * T * (size_t)ncounters + i;
    uint32_t v[SCAN_BATCH_MAX];
#pragma unroll
    for (uint32_t t = 0; t < (uint32_t)SCAN_BATCH_MAX; t++) v[t] = (t < T && ((gomask >> t) & 1u)) ? dr[(size_t)t * ncounters] : 0u;   // all loads in flight
    unsigned long long sum = 0;
#pragma unroll
    for (uint32_t t = 0; t < (uint32_t)SCAN_BATCH_MAX; t++) sum += v[t];
    if (sum) atomicAdd(&ctr[i], sum);
}
// The gathered header of a step, on every rank: each block must be sealed with this step's tag.  `fatal` (delta exchange:
// the used-bin counts of a void block cannot be trusted and the spectra behind them are wiped) raises HULK_ERR_COMM; after
// a spectra exchange the header only carries the verdicts for the next step, which a void block turns into "full" on every
// rank (hulk_comm.hip).  The kernel itself stores the header to the host's mapped copy: no copy engine between the
// exchange and the host's view of it.
__global__ __launch_bounds__(256) void k_shard_check(const uint32_t *__restrict__ hdr_all, uint32_t world, uint32_t step_tag,
                                                     DevState *st, uint32_t *__restrict__ h_out, int fatal) {
    const uint32_t n = world * SHARD_HDR;
    bool bad = false;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t v = hdr_all[i];
        if ((i % SHARD_HDR) == SHARD_TAG && v != step_tag) bad = true;
        if (h_out) __builtin_nontemporal_store(v, &h_out[i]);
    }
    if (bad && fatal) set_error(st, -37);                                      // HULK_ERR_COMM
    __threadfence_system();
}
}  // namespace

hipError_t launch_shard_check(hipStream_t s, const uint32_t *d_hdr_all, uint32_t world, uint32_t step_tag, DevState *st,
                              uint32_t *h_out, int fatal) {
    prof_mark(s, "k_shard_check");
    hipLaunchKernelGGL(k_shard_check, dim3(1), dim3(256), 0, s, d_hdr_all, world, step_tag, st, h_out, fatal);
    return hipGetLastError();
}

hipError_t launch_shard_local(hipStream_t s, uint32_t *d_hists, const uint16_t *d_pos16, uint32_t *d_hdr, uint32_t *d_delta,
                              int depth, int width, const FlushBatch &fb) {
    if (fb.count == 0) return hipSuccess;
    const int chunks = (fb.num_bins + 63) / 64;
    const int seg_chunks = (chunks + CMS_SEGS - 1) / CMS_SEGS;
    prof_mark(s, "k_shard_local");
    hipLaunchKernelGGL(k_shard_local, dim3(CMS_SEGS, fb.count), dim3(512), (size_t)depth * width * 4, s, d_hists, d_pos16, d_hdr,
                       d_delta, depth, width, seg_chunks, fb);
    return hipGetLastError();
}

hipError_t launch_shard_apply(hipStream_t s, const uint32_t *d_hdr_all, const uint32_t *d_delta_all, unsigned long long *d_ctr,
                              int depth, int width, uint32_t world, uint32_t T, uint32_t step_intervals, int32_t num_bins,
                              DevState *st, uint32_t step_tag) {
    const int nc = depth * width;
    const uint32_t ranks = std::min<uint32_t>(world, (step_intervals + T - 1) / T);      // ranks that hold intervals of this step
    prof_mark(s, "k_shard_apply");
    hipLaunchKernelGGL(k_shard_apply, dim3((nc + 255) / 256, ranks ? ranks : 1), dim3(256), 0, s, d_hdr_all, d_delta_all, d_ctr, nc,
                       world, T, step_intervals, num_bins, st, step_tag);
    return hipGetLastError();
}

}  // namespace hulk
