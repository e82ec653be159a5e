// hulk_cws.hip — consistent weighted sampling: the histosketch update, its tables, `hulk smash`.
//   k_flush_decide     whole-batch bound: can any element still change the sketch?
//   K4  k_cws_scan         fp32 pass over K = c*exp(b-r): per (interval, slot, tile) minimum, bound-pruned
//       k_cws_resolve(+_drift)/k_cws_apply   exact fp64 re-evaluation with the literal formula of
//                          src/histosketch/histosketch.go:30-33 and the slot update (histosketch.go:135-153)
//       k_alfg_*/k_rng_candidates/k_cws_eval/k_cws_scatter/k_cws_beta/k_build_k32   the CWS tables (histosketch.go:95-126)
//       k_smash            pairwise distances of `hulk smash`
#include "hulk_device.h"

#include <math.h>
#include <stdlib.h>
#include <algorithm>

namespace hulk {
namespace {

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
                                                       DevState *st, FlushBatch fb, int enable,
                                                       unsigned long long *seal, uint32_t seal_tag) {
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
    if (enable == 2) pass = false;                               // seal only (a rank without slots has nothing to protect)
    else if (!enable || m == 0) pass = true;                     // an untouched counter: estimates can be as small as 1
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
    // seal: the verdict goes to a rank's exchange header instead (hulk_step_sharded: it governs the NEXT step) — as ONE
    // 64-bit word {step tag : verdict}, the LAST store into the block (everything else of the block was written by kernels
    // in front of this one on the stream): a block with the step's tag has all of its content, a block without is void
    if (tid == 0) {
        if (seal) *seal = ((unsigned long long)seal_tag << 32) | (anypass ? 1ull : 0ull);
        else st->skip_exact[fb.parity] = anypass ? 0u : 1u;
    }
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

// Without concept drift only min_t K * rcp_t[bin] over the batch's flushed intervals is wanted (k_cws_scan<MERGE>), and a
// rounded fp32 product is monotone in either factor: for K < 0 the minimum is K * max_t rcp_t, for K >= 0 it is
// K * min_t rcp_t — so min(K * rmax, K * rmin) with two per-bin vectors equals the minimum over the T products bit for bit
// (NaN = bin not in an interval's stream is passed over by v_max / v_min; a bin in no interval stays NaN and loses).
// rmm[0][bin] = max_t, rmm[1][bin] = min_t.  Reads T x 0.8 MB (k = 21), writes 1.6 MB: ~5 us, and the scan behind it does
// 12 VALU per row of a tile instead of 8 per row AND interval (128 at T = 16) against 64 KB less L2 traffic per tile.
__global__ __launch_bounds__(256) void k_rcp_minmax(const float *__restrict__ rcp32, float *__restrict__ rmm, size_t row_stride,
                                                    const DevState *st, FlushBatch fb) {
    if (st->skip_exact[fb.parity]) return;
    const size_t col = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (col >= row_stride) return;
    const uint32_t gomask = batch_gomask(st, fb);
    const float qnan = __builtin_nanf("");
    floatx4 hi = (floatx4)(qnan), lo = (floatx4)(qnan);
    for (int t = 0; t < (int)fb.count; t++) {
        if (!((gomask >> t) & 1u)) continue;
        const floatx4 v = *(const floatx4 *)(rcp32 + (size_t)t * row_stride + col);
        hi.x = fmaxf(hi.x, v.x); hi.y = fmaxf(hi.y, v.y); hi.z = fmaxf(hi.z, v.z); hi.w = fmaxf(hi.w, v.w);
        lo.x = fminf(lo.x, v.x); lo.y = fminf(lo.y, v.y); lo.z = fminf(lo.z, v.z); lo.w = fminf(lo.w, v.w);
    }
    *(floatx4 *)(rmm + col) = hi;
    *(floatx4 *)(rmm + row_stride + col) = lo;
}

// ------------------------------------------------------------------------------------------
// K4a: the HBM-bound pass.  A[t][slot][bin] = K[slot][bin] * (1/f_t[bin]); minimum per
// (interval t, slot, 256-bin wave tile).  A wave streams SCAN_ROWS rows x 256 bins of K exactly once (16 B per lane per
// row, SCAN_ROWS independent loads in flight) and re-uses the registers for every interval of the batch, so the table
// is read once per BATCH, not per interval.
//
// Branch and bound first (k_scan_test): AddElement only ever replaces a slot's weight by a SMALLER A, and over a wave
// tile A = K * (1/f) >= min(K) * max(1/f) (min(K) < 0; min(K) * min(1/f) otherwise).  A wave tile whose bound cannot
// get below the current weight of any of its 8 slots for any interval of the batch is not read at all — after the first
// intervals of a stream that is nearly every tile, because count-min estimates only grow.  The weights at batch start
// are used (they only fall during the batch), with the same 1e-5 relative band the fp64 resolve uses around fp32 values.
// The verdicts are ONE BIT per (slot group, wave tile): scanmap[group][tile / 64].  k_cws_scan and everything behind it
// (k_slot_tmin, k_cws_resolve, k_cws_resolve_drift) only touch tiles whose bit is set; tilemin entries of other tiles are
// never written nor read.  (The first version tested inside the scan, one wave per tile, and wrote +inf for every
// pruned tile: at k = 31, sketchSize 1024 that was 462 k waves and 236 MB of +inf per flush — 350 us to read 33 tiles.)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scan_test(const float *__restrict__ kmin32, const float *__restrict__ rext,
                                                   const double *__restrict__ weights, int slot_begin, int slots,
                                                   int wtiles, int wwords, double drift_dw, const DevState *st,
                                                   FlushBatch fb, unsigned long long *__restrict__ scanmap,
                                                   unsigned long long *__restrict__ visited, uint32_t *__restrict__ scanlist,
                                                   uint32_t *__restrict__ scanlist_n) {
    if (st->skip_exact[fb.parity]) return;                       // k_flush_decide: nothing in this batch can matter
    // workgroup = one column word (64 wave tiles) x four slot groups, the groups running fastest over the grid: the list below
    // comes out column by column, so the waves that scan neighbouring items share a column's reciprocal vectors
    const int grp = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int word = blockIdx.y;
    if (grp >= (slots + SCAN_ROWS - 1) / SCAN_ROWS) return;
    const int wt = word * 64 + lane;
    const uint32_t gomask = batch_gomask(st, fb);
    bool pass = false;
    if (wt < wtiles) {
        if (!kmin32) pass = true;                                // pruning off: every tile is read
        else {
            float rmax[SCAN_BATCH_MAX], rmin[SCAN_BATCH_MAX];
#pragma unroll
            for (int t = 0; t < SCAN_BATCH_MAX; t++) {
                rmax[t] = 0.f; rmin[t] = 0.f;
                if (t < (int)fb.count && ((gomask >> t) & 1u)) {
                    const floatx2 v = *(const floatx2 *)(rext + ((size_t)t * wtiles + wt) * 2);
                    rmax[t] = v.x; rmin[t] = v.y;
                }
            }
            for (int row = 0; row < SCAN_ROWS; row++) {
                const int slot = grp * SCAN_ROWS + row;
                if (slot >= slots) break;
                const double km = (double)kmin32[(size_t)slot * wtiles + wt];
                double w = weights[slot_begin + slot];
                // concept drift (drift_dw = decayWeight > 0): the update test is A < w / decayWeight and w may move either
                // way — but a NEGATIVE weight can only be replaced by a smaller one (A < w/dw < w), so its threshold of
                // the whole batch is at most w_start / dw; a slot whose weight is not negative is simply never pruned
                if (drift_dw > 0.0) { if (w < 0.0) w = w / drift_dw; else pass = true; }
                const double thr = w + 1e-5 * fabs(w) + 1e-37;
#pragma unroll
                for (int t = 0; t < SCAN_BATCH_MAX; t++) {
                    if (!(rmax[t] > 0.f)) continue;              // interval not flushed, or none of its elements falls into the tile
                    const double bound = km < 0.0 ? km * (double)rmax[t] : km * (double)rmin[t];
                    if (bound <= thr) pass = true;
                }
            }
        }
    }
    const unsigned long long mask = __ballot(pass);
    if (lane == 0) {
        scanmap[(size_t)grp * wwords + word] = mask;
        if (mask) atomicAdd(&visited[(blockIdx.x + blockIdx.y * gridDim.x) & (MIN_SLOTS - 1)], (unsigned long long)__popcll(mask));
    }
    // ... and the tiles to read as a LIST (slot group << 12 | wave tile; wave tiles < 4096 for every num_bins <= 2^20): the scan
    // runs over it with a fixed grid.  (One workgroup of 16 waves per 16 tiles of the whole table, most of them returning at
    // once, was 235 us per flush at k = 31, sketchSize 1024 for the 3 % of the tiles the bounds let through: 30 k workgroups.)
    if (mask && scanlist) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(scanlist_n, (uint32_t)__popcll(mask));
        base = (uint32_t)__shfl((int)base, 0);
        if (pass) scanlist[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))] = ((uint32_t)grp << 12) | (uint32_t)wt;
    }
}

// workgroup of 16 waves = (quarter of a column word = 16 wave tiles, slot group); wave w takes wave tile 64 cw + 16 quarter + w
// MERGE (no concept drift): AddElement's update is then a running minimum with strict <, so what a batch leaves in a
// slot is the smallest A over ALL its (interval, bin) pairs, the earliest pair winning ties (histosketch.go:139-153) — the
// per-interval minima are not needed.  The lanes keep ONE running minimum per row over the batch's intervals and the wave
// reduces it once per tile instead of once per (tile, interval): the 19-instruction transposed reduction was 36 % of the
// issue cycles of an interval's work on a tile (235 -> 151 cycles; profiles/r04_scan.txt).  k_cws_resolve<true> then
// re-evaluates the candidate tiles for every interval of the batch and orders the exact values by (A, interval, bin).
// MODE 0: per-interval minima (concept drift); 1: MERGE over the T reciprocal vectors (kept as the comparator of MODE 2:
// HULK_SCAN_MERGE_LOOP in the profiling build); 2: MERGE over the two vectors of k_rcp_minmax (`rcp32` = rmm) — the product path.
// One wave tile (8 rows x 256 bins of K) by one wave:
template <int MODE>
__device__ __forceinline__ void scan_wave_tile(const float *__restrict__ k32, const float *__restrict__ rcp32, float *__restrict__ tilemin,
                                               int slots, int ngroups, int wtiles, size_t row_stride, const FlushBatch &fb, uint32_t gomask,
                                               int grp, int wt, int lane) {
    constexpr bool MERGE = MODE != 0;
        const size_t col = (size_t)wt * 256 + (size_t)lane * 4;
        floatx4 kv[SCAN_ROWS];
#pragma unroll
        for (int r = 0; r < SCAN_ROWS; r++) {
            const int slot = grp * SCAN_ROWS + r;
            if (slot < slots)
                kv[r] = __builtin_nontemporal_load((const floatx4 *)(k32 + (size_t)slot * row_stride + col));
            else
                kv[r] = (floatx4)(0.f);
        }
        if (MODE == 2) {
            // min_t K * rcp_t = min(K * max_t rcp_t, K * min_t rcp_t): 8 v_mul + 4 v_min3 per row, whatever the batch size
            const floatx4 hi = *(const floatx4 *)(rcp32 + col), lo = *(const floatx4 *)(rcp32 + row_stride + col);
            float m[SCAN_ROWS];
#pragma unroll
            for (int r = 0; r < SCAN_ROWS; r++) {
                const float a = fminf(fminf(kv[r].x * hi.x, kv[r].x * lo.x), kv[r].y * hi.y);
                const float b = fminf(fminf(kv[r].y * lo.y, kv[r].z * hi.z), kv[r].z * lo.z);
                m[r] = fminf(fminf(fminf(a, b), fminf(kv[r].w * hi.w, kv[r].w * lo.w)), INFINITY);
            }
            static_assert(SCAN_ROWS == 8, "wave_min8_by_row reduces exactly 8 rows");
            const float mine = wave_min8_by_row(m);
            if ((lane & 7) == 0)                                    // plane 0 of tilemin: [slot group][wave tile][row]
                tilemin[((size_t)grp * wtiles + (size_t)wt) * SCAN_ROWS + (lane >> 3)] = mine;
            return;
        }
        floatx4 rc_next = *(const floatx4 *)(rcp32 + col);
        float acc[SCAN_ROWS];
#pragma unroll
        for (int r = 0; r < SCAN_ROWS; r++) acc[r] = INFINITY;
        for (int t = 0; t < (int)fb.count; t++) {
            const floatx4 rc = rc_next;
            if (t + 1 < (int)fb.count) rc_next = *(const floatx4 *)(rcp32 + (size_t)(t + 1) * row_stride + col);
            if (!((gomask >> t) & 1u)) continue;
            if (MERGE) {
                // NaN (bin not in the stream) loses every v_min; the running minimum starts at +inf
#pragma unroll
                for (int r = 0; r < SCAN_ROWS; r++)
                    acc[r] = fminf(fminf(fminf(fminf(kv[r].x * rc.x, kv[r].y * rc.y), kv[r].z * rc.z), kv[r].w * rc.w), acc[r]);
                continue;
            }
            float m[SCAN_ROWS];
            // v_mul_f32 x4 + v_min3_f32 x2 per row.  (v_pk_mul_f32 halves the multiplies but runs this loop 2x
            // SLOWER on MI355X — measured 304 vs 150 us — so the products stay scalar.)  NaN (bin not in the
            // stream) loses every v_min; INFINITY keeps an all-NaN lane out of the reduction.
#pragma unroll
            for (int r = 0; r < SCAN_ROWS; r++)
                m[r] = fminf(fminf(fminf(fminf(kv[r].x * rc.x, kv[r].y * rc.y), kv[r].z * rc.z), kv[r].w * rc.w), INFINITY);
            static_assert(SCAN_ROWS == 8, "wave_min8_by_row reduces exactly 8 rows");
            const float mine = wave_min8_by_row(m);                // lane l: minimum of row l / 8 over the wave
            // tilemin[t][slot group][wave tile][row]: 8 lanes write the 8 rows of a wave tile (32 contiguous bytes)
            if ((lane & 7) == 0)
                tilemin[(((size_t)t * ngroups + grp) * wtiles + (size_t)wt) * SCAN_ROWS + (lane >> 3)] = mine;
        }
        if (MERGE) {
            static_assert(SCAN_ROWS == 8, "wave_min8_by_row reduces exactly 8 rows");
            const float mine = wave_min8_by_row(acc);             // lane l: minimum of row l / 8 over the wave and the batch
            if ((lane & 7) == 0)                                    // plane 0 of tilemin: [slot group][wave tile][row]
                tilemin[((size_t)grp * wtiles + (size_t)wt) * SCAN_ROWS + (lane >> 3)] = mine;
        }
}

// the scan over the list of k_scan_test: a fixed grid of 4-wave workgroups, wave w takes items w, w + waves, ...
template <int MODE>
__global__ __launch_bounds__(256) void k_cws_scan_list(const float *__restrict__ k32, const float *__restrict__ rcp32,
                                                       float *__restrict__ tilemin, int slots, int ntiles, size_t row_stride,
                                                       const DevState *st, FlushBatch fb, const uint32_t *__restrict__ scanlist,
                                                       const uint32_t *__restrict__ scanlist_n) {
    if (st->skip_exact[fb.parity]) return;
    const uint32_t n = *scanlist_n;
    const int lane = threadIdx.x & 63;
    // XCD-aware (workgroup b runs on XCD b % 8; the grid is a multiple of 8): every XCD takes one contiguous eighth of the
    // list — the list is column-major, so the slot groups that read the same column's reciprocal vectors meet in one L2
    const uint32_t xcd = blockIdx.x & 7u, per = (n + 7u) / 8u, first = xcd * per, last = first + per < n ? first + per : n;
    const uint32_t wave = (blockIdx.x >> 3) * 4 + (threadIdx.x >> 6), waves = (gridDim.x >> 3) * 4;
    if (first + wave >= last) return;
    const int ngroups = (slots + SCAN_ROWS - 1) / SCAN_ROWS, wtiles = ntiles * 4;
    const uint32_t gomask = batch_gomask(st, fb);
    for (uint32_t i = first + wave; i < last; i += waves) {
        const uint32_t item = scanlist[i];
        scan_wave_tile<MODE>(k32, rcp32, tilemin, slots, ngroups, wtiles, row_stride, fb, gomask, (int)(item >> 12), (int)(item & 4095u), lane);
    }
}

// (the grid form: one workgroup of 16 waves per 16 wave tiles of the whole table — HULK_SCAN_GRID in the profiling build)
template <int MODE>
__global__ __launch_bounds__(1024) void k_cws_scan(const float *__restrict__ k32,
                                                   const float *__restrict__ rcp32,
                                                   float *__restrict__ tilemin, int slots, int ntiles,
                                                   size_t row_stride, const DevState *st, FlushBatch fb,
                                                   const unsigned long long *__restrict__ scanmap, int wwords) {
    // XCD-aware order (workgroup b lands on XCD b % 8): XCD x works through column unit 8*chunk + x (16 wave tiles) for
    // ALL slot groups before moving on; the 8 XCDs stream 8 adjacent 16 KB pieces of the same K rows at the same time.
    if (st->skip_exact[fb.parity]) return;
    const int ngroups = (slots + SCAN_ROWS - 1) / SCAN_ROWS;
    const int r = (int)(blockIdx.x >> 3), grp = r % ngroups, unit = (r / ngroups) * 8 + (int)(blockIdx.x & 7);
    const int cw = unit >> 2, quarter = unit & 3;                // unit = 16 wave tiles: fine enough to balance the 8 XCDs
    if (cw >= wwords) return;
    const unsigned long long mask = scanmap[(size_t)grp * wwords + cw];
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    if (!((mask >> (16 * quarter + wid)) & 1ull)) return;        // wave-uniform: this tile cannot change the sketch
    scan_wave_tile<MODE>(k32, rcp32, tilemin, slots, ngroups, ntiles * 4, row_stride, fb, batch_gomask(st, fb), grp,
                         cw * 64 + quarter * 16 + wid, lane);
}

// ------------------------------------------------------------------------------------------
// K4b: per slot, interval by interval: re-evaluate in fp64 — with the literal getSample formula —
// every wave tile whose fp32 minimum is within a relative band of the slot's fp32 minimum, then
// apply AddElement's update rule.  The band (1e-5 rel + 1e-37 abs) is >30x the worst fp32 error
// of K4a, so the true fp64 argmin (and every exact tie, for earliest-wins) is always inside a
// re-evaluated tile.
// ------------------------------------------------------------------------------------------
constexpr int WTILE = SCAN_TILE / 4;
// MERGE (see k_cws_scan): one workgroup per slot; tilemin holds the batch's minima, the candidate tiles are re-evaluated
// for every flushed interval of the batch and the exact values ordered by (A, interval, bin) — the stream order of ties.
template <bool MERGE>
__global__ __launch_bounds__(256) void k_cws_resolve(const double *__restrict__ rcb,
                                                     const double *__restrict__ f64,
                                                     const float *__restrict__ tilemin,
                                                     double *__restrict__ candA, int32_t *__restrict__ candB,
                                                     int slots, int ntiles, const unsigned long long *__restrict__ scanmap,
                                                     int wwords, const DevState *st, FlushBatch fb) {
    extern __shared__ __align__(16) unsigned char smem[];
    float *tm = (float *)smem;                       // [wtiles]
    __shared__ float redf[4];
    __shared__ double redA[4];
    __shared__ long long redK[4];
    __shared__ int ncand;
    __shared__ int cand[64];
    if (st->skip_exact[fb.parity]) return;
    const int slot = blockIdx.x, t = blockIdx.y;     // local slot, interval of the batch (MERGE: t = 0, the whole batch)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wtiles = ntiles * 4;
    const int32_t num_bins = fb.num_bins;
    double bestA = INFINITY; int32_t bestB = 0x7fffffff; int bestT = 0x7fffffff;
    const uint32_t gomask = MERGE ? batch_gomask(st, fb) : 0u;   // (whole waves: every thread of the block is active here)
    if (MERGE ? gomask != 0u : flush_go(st, fb, t)) {  // block-uniform
        const double *row = rcb + (size_t)slot * (size_t)num_bins * 3;
        const int ngroups = (slots + SCAN_ROWS - 1) / SCAN_ROWS;
        const float *tmin_t = tilemin + (((size_t)t * ngroups + slot / SCAN_ROWS) * wtiles) * SCAN_ROWS + (slot % SCAN_ROWS);
        const double *ft = f64 + (size_t)t * (size_t)num_bins;
        if (tid == 0) ncand = 0;
        const unsigned long long *smap = scanmap + (size_t)(slot / SCAN_ROWS) * wwords;   // tiles the scan read
        float g = INFINITY;
        for (int x = tid; x < wtiles; x += blockDim.x) {
            const float v = ((smap[x >> 6] >> (x & 63)) & 1ull) ? tmin_t[(size_t)x * SCAN_ROWS] : INFINITY;
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
            if (MERGE) {
                // every candidate tile (all tiles inside the band when there are more than 64), every flushed interval
                for (int ci = 0; ci < (nc <= 64 ? nc : wtiles); ci++) {
                    int x = ci;
                    if (nc <= 64) x = cand[ci]; else if (!(tm[ci] <= thr)) continue;
                    const int32_t bin = x * WTILE + tid;             // WTILE == blockDim.x
                    if (bin >= num_bins) continue;
                    const double r = row[(size_t)bin * 3 + 0];
                    const double c = row[(size_t)bin * 3 + 1];
                    const double b = row[(size_t)bin * 3 + 2];
                    const double er = exp(r);
                    for (int tt = 0; tt < (int)fb.count; tt++) {
                        if (!((gomask >> tt) & 1u)) continue;
                        const double f = f64[(size_t)tt * (size_t)num_bins + bin];
                        if (f == 0.0) continue;
                        const double Yka = exp(log(f) - b);
                        const double A = c / (Yka * er);
                        if (A < bestA || (A == bestA && (tt < bestT || (tt == bestT && bin < bestB)))) { bestA = A; bestB = bin; bestT = tt; }
                    }
                }
            } else if (nc <= 64) {
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
            // (interval, bin) as one key: the order of the stream
            long long bestK = ((long long)bestT << 32) | (long long)(uint32_t)bestB;
            for (int off = 32; off; off >>= 1) {
                const double oA = __shfl_xor(bestA, off);
                const long long oK = __shfl_xor(bestK, off);
                if (oA < bestA || (oA == bestA && oK < bestK)) { bestA = oA; bestK = oK; }
            }
            if (lane == 0) { redA[wid] = bestA; redK[wid] = bestK; }
            __syncthreads();
#pragma unroll
            for (int x = 0; x < 4; x++)
                if (redA[x] < bestA || (redA[x] == bestA && redK[x] < bestK)) { bestA = redA[x]; bestK = redK[x]; }
            bestB = (int32_t)(uint32_t)bestK;
        }
    }
    if (tid == 0) { candA[(size_t)t * slots + slot] = bestA; candB[(size_t)t * slots + slot] = bestB; }
}

// AddElement's slot update (histosketch.go:150-153, no drift) in interval order: the element with
// the smallest A of interval t replaces the slot iff it is strictly below the running weight.
__global__ void k_cws_apply(const double *__restrict__ candA, const int32_t *__restrict__ candB,
                            unsigned long long *__restrict__ mins, double *__restrict__ weights,
                            int slots, int slot_begin, const DevState *st, FlushBatch fb, int merged) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= slots || st->skip_exact[fb.parity]) return;
    const int gs = slot_begin + slot;
    double w = weights[gs]; unsigned long long m = mins[gs];
    for (int t = 0; t < (merged ? 1 : (int)fb.count); t++) {
        if (!merged && !flush_go(st, fb, t)) continue;          // (merged: plane 0 is the batch's one candidate, or none)
        const double A = candA[(size_t)t * slots + slot];
        const int32_t b = candB[(size_t)t * slots + slot];
        if (b != 0x7fffffff && A < w) { w = A; m = (unsigned long long)b; }
    }
    weights[gs] = w; mins[gs] = m;
}

// Minimum of a slot's tile minima per interval: slotmin[t][slot] = min over the wave tiles of tilemin[t][group][tile][row].
// One coalesced pass over tilemin, so that k_cws_resolve_drift can pass over the (slot, interval) pairs that hold no
// candidate without touching their rows (it used to stage every row — 8-float strides — behind two barriers per interval).
__global__ __launch_bounds__(256) void k_slot_tmin(const float *__restrict__ tilemin, float *__restrict__ slotmin, int wtiles,
                                                   int ngroups, const unsigned long long *__restrict__ scanmap, int wwords,
                                                   const DevState *st, FlushBatch fb) {
    __shared__ float red[4][8];
    const int grp = blockIdx.x, t = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const floatx4 *src = (const floatx4 *)(tilemin + (((size_t)t * ngroups + grp) * wtiles) * SCAN_ROWS);
    const unsigned long long *smap = scanmap + (size_t)grp * wwords;
    floatx4 lo = (floatx4)(INFINITY), hi = (floatx4)(INFINITY);   // rows 0..3 / 4..7
    for (int x = tid; x < wtiles; x += 256) {
        if (!((smap[x >> 6] >> (x & 63)) & 1ull)) continue;       // the scan did not read this tile
        const floatx4 a = src[2 * x], b = src[2 * x + 1];
        lo.x = fminf(lo.x, a.x); lo.y = fminf(lo.y, a.y); lo.z = fminf(lo.z, a.z); lo.w = fminf(lo.w, a.w);
        hi.x = fminf(hi.x, b.x); hi.y = fminf(hi.y, b.y); hi.z = fminf(hi.z, b.z); hi.w = fminf(hi.w, b.w);
    }
    float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int r = 0; r < 8; r++) {
        for (int off = 32; off; off >>= 1) v[r] = fminf(v[r], __shfl_xor(v[r], off));
        if (lane == 0) red[wid][r] = v[r];
    }
    __syncthreads();
    if (tid < 8) slotmin[((size_t)t * ngroups + grp) * SCAN_ROWS + tid] = fminf(fminf(red[0][tid], red[1][tid]), fminf(red[2][tid], red[3][tid]));
    (void)st;
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
                                                           const float *__restrict__ slotmin,
                                                           const unsigned long long *__restrict__ scanmap, int wwords,
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
        {   // no tile of this interval can hold a trigger for the current threshold (same screen as below): next interval
            const double thr0 = w / decay_weight;
            if (thr0 != thr0) continue;
            const double lim0 = thr0 + 1e-5 * fabs(thr0) + 1e-37;
            if (lim0 < INFINITY && !((double)slotmin[(size_t)t * ngroups * SCAN_ROWS + slot] <= lim0)) continue;   // block-uniform
        }
        __syncthreads();
        const float *tmin_t = tilemin + (((size_t)t * ngroups + slot / SCAN_ROWS) * wtiles) * SCAN_ROWS + (slot % SCAN_ROWS);
        const double *ft = f64 + (size_t)t * (size_t)num_bins;
        {
            const unsigned long long *smap = scanmap + (size_t)(slot / SCAN_ROWS) * wwords;       // tiles the scan read
            for (int x = tid; x < wtiles; x += blockDim.x)
                tm[x] = ((smap[x >> 6] >> (x & 63)) & 1ull) ? tmin_t[(size_t)x * SCAN_ROWS] : INFINITY;
        }
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
                                                   uint32_t first_chunk, uint32_t n_chunks, uint32_t stride,
                                                   uint32_t span, uint32_t steps) {
    // workgroup b starts from windows[first_chunk + b * span] (valid) and takes `steps` jumps of `stride` chunks with
    // the polynomial `coef` (= x^(stride * 2^20)), storing every window it reaches — as far as the chunks go.
    // Two levels: ONE workgroup walks every 64th chunk with the far polynomial, then one workgroup per 64 chunks fills
    // in the 63 between (a single walk over all chunks was 130 ms at k = 31, sketchSize 1024: 5.5 k dependent jumps).
    __shared__ uint64_t E[1216], C[608];
    const int tid = threadIdx.x;
    const uint32_t last = first_chunk + n_chunks;                 // one past the last chunk
    uint32_t c = first_chunk + blockIdx.x * span;
    if (c >= last) return;
    if (tid < 607) { C[tid] = coef[tid]; E[tid] = windows[(size_t)c * 607 + tid]; }
    __syncthreads();
    for (uint32_t i = 0; i < steps && c + stride < last; i++, c += stride) {
        for (int base = 607; base < 1213; base += 273) {          // extend the window by 606 values, 273 at a time
            const int j = base + tid;
            if (tid < 273 && j < 1213) E[j] = E[j - 607] + E[j - 273];
            __syncthreads();
        }
        uint64_t acc = 0;
        if (tid < 607) for (int x = 0; x < 607; x++) acc += C[x] * E[x + tid];
        __syncthreads();
        if (tid < 607) { E[tid] = acc; windows[(size_t)(c + stride) * 607 + tid] = acc; }
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
// Every pair (s, q) accumulates over the slots IN ORDER, so the fp64 sums are bit-identical to the Go loops.
//   k_smash_prep  once per call: mins -> float64 (the reference compares them as float64, sketchio.go:271-277), weights ->
//                 |w| (max(max(w,0), max(-w,0)), NaN stays NaN), both stored SLOT-major ([slot][sketch]) so that a tile's rows
//                 are contiguous
//   k_smash       register tile: a thread owns 4 subjects x 4 queries, a workgroup of 128 threads a tile of 32 subjects x 64
//                 queries; chunks of 32 slots go through LDS as [slot][row] (the prep's layout: 16-byte loads in, 16-byte LDS
//                 stores, no transposition) in two buffers — chunk n+1 is loaded into registers before chunk n is computed
//                 and stored behind it, one barrier per chunk.  Per slot a thread reads its 4 subject mins, 4 subject
//                 weights and 4 query mins with six ds_read_b128 (0.375 LDS reads per (pair, slot), was 3) and runs 16
//                 independent accumulators; `+= equal ? |w| : 0.0` is the reference's conditional add bit for bit (the sums
//                 are non-negative: x + 0.0 == x); the union of the weighted metric is the sum of the subject's |w| whatever
//                 the query (both branches of distances.go:58-68 add max(wA, wB) = |w| when hsB = subject).
// ==========================================================================================
constexpr int SMASH_TS = 32, SMASH_TQ = 64, SMASH_CH = 32, SMASH_PAD = 2;
__global__ __launch_bounds__(256) void k_smash_prep(const unsigned long long *__restrict__ mins, const double *__restrict__ weights,
                                                    uint32_t N, uint32_t S, uint32_t NP, double *__restrict__ mT, double *__restrict__ wT) {
    // 32 x 32 tiles through LDS: reads run along the slots of a sketch, writes along the sketches of a slot
    __shared__ double tm[32][33], tw[32][33];
    const uint32_t n0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (uint32_t r = ty; r < 32; r += 8) {
        const uint32_t n = n0 + r, c = c0 + tx;
        const bool ok = n < N && c < S;
        tm[r][tx] = ok ? (double)mins[(size_t)n * S + c] : 0.0;
        tw[r][tx] = ok ? fabs(weights[(size_t)n * S + c]) : 0.0;
    }
    __syncthreads();
    for (uint32_t r = ty; r < 32; r += 8) {
        const uint32_t c = c0 + r, n = n0 + tx;
        if (c < S && n < NP) { mT[(size_t)c * NP + n] = tm[tx][r]; wT[(size_t)c * NP + n] = tw[tx][r]; }
    }
}

template <int METRIC>
__global__ __launch_bounds__(128) void k_smash(const double *__restrict__ mT, const double *__restrict__ wT, uint32_t N, uint32_t NP,
                                               uint32_t S, double *__restrict__ out) {
    __shared__ __align__(16) double ma[2][SMASH_CH][SMASH_TS + SMASH_PAD], wa[2][SMASH_CH][SMASH_TS + SMASH_PAD];
    __shared__ __align__(16) double mb[2][SMASH_CH][SMASH_TQ + SMASH_PAD];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;     // query quad, subject quad inside the tile
    const uint32_t s0 = blockIdx.y * SMASH_TS, q0 = blockIdx.x * SMASH_TQ;
    // staging: thread t moves slot (t / 4) of the chunk: 8 subject rows (mins, weights) and 16 query rows from (t % 4) on
    const int lc = tid >> 2, lr = tid & 3;
    double acc[4][4], uni[4];
    uint32_t cnt[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uni[i] = 0.0;
#pragma unroll
        for (int j = 0; j < 4; j++) { acc[i][j] = 0.0; cnt[i][j] = 0; }
    }
    double2 ra[4], rw[4], rb[8];
    auto fetch = [&](uint32_t c0) {
        const uint32_t col = c0 + (uint32_t)lc;
        const bool ok = col < S;                                    // (rows past N hold zeros: NP is N rounded up to the tile)
        const double2 *pa = (const double2 *)(mT + (size_t)col * NP + s0 + 8 * lr);
        const double2 *pw = (const double2 *)(wT + (size_t)col * NP + s0 + 8 * lr);
        const double2 *pb = (const double2 *)(mT + (size_t)col * NP + q0 + 16 * lr);
#pragma unroll
        for (int x = 0; x < 4; x++) { ra[x] = ok ? pa[x] : make_double2(0.0, 0.0); if (METRIC == 1) rw[x] = ok ? pw[x] : make_double2(0.0, 0.0); }
#pragma unroll
        for (int x = 0; x < 8; x++) rb[x] = ok ? pb[x] : make_double2(0.0, 0.0);
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int x = 0; x < 4; x++) { *(double2 *)&ma[buf][lc][8 * lr + 2 * x] = ra[x]; if (METRIC == 1) *(double2 *)&wa[buf][lc][8 * lr + 2 * x] = rw[x]; }
#pragma unroll
        for (int x = 0; x < 8; x++) *(double2 *)&mb[buf][lc][16 * lr + 2 * x] = rb[x];
    };
    fetch(0);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (uint32_t c0 = 0; c0 < S; c0 += SMASH_CH, buf ^= 1) {
        const bool more = c0 + SMASH_CH < S;
        if (more) fetch(c0 + SMASH_CH);                             // in flight under this chunk's arithmetic
        const uint32_t lim = S - c0 < (uint32_t)SMASH_CH ? S - c0 : (uint32_t)SMASH_CH;
#pragma unroll 2
        for (uint32_t c = 0; c < lim; c++) {                       // (unrolled by two: the next slot's six LDS reads are in flight under this one's 72 VALU)
            const double2 a01 = *(const double2 *)&ma[buf][c][4 * ty], a23 = *(const double2 *)&ma[buf][c][4 * ty + 2];
            const double2 b01 = *(const double2 *)&mb[buf][c][4 * tx], b23 = *(const double2 *)&mb[buf][c][4 * tx + 2];
            const double a[4] = {a01.x, a01.y, a23.x, a23.y}, b[4] = {b01.x, b01.y, b23.x, b23.y};
            if (METRIC == 1) {
                const double2 w01 = *(const double2 *)&wa[buf][c][4 * ty], w23 = *(const double2 *)&wa[buf][c][4 * ty + 2];
                const double w[4] = {w01.x, w01.y, w23.x, w23.y};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    uni[i] += w[i];
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[i][j] += (a[i] == b[j]) ? w[i] : 0.0;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) cnt[i][j] += (a[i] == b[j]) ? 1u : 0u;      // a count of 1.0s is exact in fp64
            }
        }
        if (more) stash(buf ^ 1);                                   // (the other buffer: nobody reads it during this chunk)
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t s = s0 + 4 * ty + i;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t q = q0 + 4 * tx + j;
            if (s < N && q < N)
                out[(size_t)s * N + q] = METRIC == 1 ? 1 - (acc[i][j] / uni[i]) : 1.0 - ((double)cnt[i][j] / (double)S);
        }
    }
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

}  // namespace

// ---------------------------------------------------------------------------- host wrappers
// A/B aid: HULK_SCAN_PER_INTERVAL=1 keeps the per-interval minima (and resolve) without concept drift too
static bool scan_merge_off() { static const bool v = HULK_EXP_ENV("HULK_SCAN_PER_INTERVAL") != nullptr; return v; }
static bool scan_merge_loop() { static const bool v = HULK_EXP_ENV("HULK_SCAN_MERGE_LOOP") != nullptr; return v; }

static bool scan_grid_form() { static const bool v = HULK_EXP_ENV("HULK_SCAN_GRID") != nullptr; return v; }
hipError_t launch_cws_scan(hipStream_t s, const float *d_k32, const float *d_rcp32, float *d_tilemin,
                           int slots, int ntiles, size_t row_stride, DevState *st, const FlushBatch &fb,
                           const float *d_kmin32, float *d_rext, const double *d_weights, int slot_begin,
                           unsigned long long *d_visited, double drift_dw, unsigned long long *d_scanmap, bool per_interval,
                           float *d_rmm, uint32_t *d_scanlist, uint32_t *d_scanlist_n, hipEvent_t scan_begin, hipEvent_t scan_end) {
    const int groups = (slots + SCAN_ROWS - 1) / SCAN_ROWS;
    const int wtiles = ntiles * 4, wwords = (wtiles + 63) / 64;
    const bool grid_form = scan_grid_form();
    if (d_kmin32) {
        prof_mark(s, "k_rcp_extrema");
        hipLaunchKernelGGL(k_rcp_extrema, dim3(ntiles, fb.count), dim3(256), 0, s, d_rcp32, d_rext, ntiles, row_stride, st, fb);
    }
    if (!grid_form) { const hipError_t e = hipMemsetAsync(d_scanlist_n, 0, 4, s); if (e != hipSuccess) return e; }
    prof_mark(s, "k_scan_test");
    hipLaunchKernelGGL(k_scan_test, dim3((groups + 3) / 4, wwords), dim3(256), 0, s, d_kmin32, d_rext, d_weights, slot_begin,
                       slots, wtiles, wwords, drift_dw, st, fb, d_scanmap, d_visited, grid_form ? nullptr : d_scanlist, d_scanlist_n);
    const int mode = (per_interval || scan_merge_off()) ? 0 : scan_merge_loop() ? 1 : 2;   // 0: concept drift, the elements are taken in stream order
    if (mode == 2) {
        prof_mark(s, "k_rcp_minmax");
        hipLaunchKernelGGL(k_rcp_minmax, dim3((unsigned)((row_stride / 4 + 255) / 256)), dim3(256), 0, s, d_rcp32, d_rmm, row_stride, st, fb);
    }
    const float *rv = mode == 2 ? d_rmm : d_rcp32;
    if (scan_begin) { const hipError_t e = hipEventRecord(scan_begin, s); if (e != hipSuccess) return e; }   // bench.py: the scan kernel alone
    if (grid_form) {
        const dim3 g((unsigned)(((wwords * 4 + 7) / 8) * 8 * groups));            // units of 16 wave tiles, 8 (one per XCD) side by side
        if (mode == 0) { prof_mark(s, "k_cws_scan"); hipLaunchKernelGGL(k_cws_scan<0>, g, dim3(1024), 0, s, d_k32, rv, d_tilemin, slots, ntiles, row_stride, st, fb, d_scanmap, wwords); }
        else if (mode == 1) { prof_mark(s, "k_cws_scan"); hipLaunchKernelGGL(k_cws_scan<1>, g, dim3(1024), 0, s, d_k32, rv, d_tilemin, slots, ntiles, row_stride, st, fb, d_scanmap, wwords); }
        else { prof_mark(s, "k_cws_scan"); hipLaunchKernelGGL(k_cws_scan<2>, g, dim3(1024), 0, s, d_k32, rv, d_tilemin, slots, ntiles, row_stride, st, fb, d_scanmap, wwords); }
    } else {
        // enough waves for the whole table to be in flight at 8 KB per wave; a short list leaves most of them nothing to do
        const size_t items = (size_t)groups * wtiles;
        const dim3 g((unsigned)((std::min<size_t>(2048, (items + 3) / 4) + 7) / 8 * 8));
        if (mode == 0) { prof_mark(s, "k_cws_scan_list"); hipLaunchKernelGGL(k_cws_scan_list<0>, g, dim3(256), 0, s, d_k32, rv, d_tilemin, slots, ntiles, row_stride, st, fb, d_scanlist, d_scanlist_n); }
        else if (mode == 1) { prof_mark(s, "k_cws_scan_list"); hipLaunchKernelGGL(k_cws_scan_list<1>, g, dim3(256), 0, s, d_k32, rv, d_tilemin, slots, ntiles, row_stride, st, fb, d_scanlist, d_scanlist_n); }
        else { prof_mark(s, "k_cws_scan_list"); hipLaunchKernelGGL(k_cws_scan_list<2>, g, dim3(256), 0, s, d_k32, rv, d_tilemin, slots, ntiles, row_stride, st, fb, d_scanlist, d_scanlist_n); }
    }
    if (scan_end) { const hipError_t e = hipEventRecord(scan_end, s); if (e != hipSuccess) return e; }
    return hipGetLastError();
}

hipError_t launch_slot_kmin(hipStream_t s, const float *d_kmin32, float *d_kminslot, int slots, int ntiles) {
    hipLaunchKernelGGL(k_slot_kmin, dim3(slots), dim3(256), 0, s, d_kmin32, d_kminslot, ntiles * 4);
    return hipGetLastError();
}

hipError_t launch_flush_decide(hipStream_t s, const unsigned long long *d_ctr, int ncounters, const float *d_kminslot,
                               const double *d_weights, int slots, int slot_begin, DevState *st, const FlushBatch &fb,
                               int enable, unsigned long long *d_seal, uint32_t seal_tag) {
    prof_mark(s, "k_flush_decide");
    hipLaunchKernelGGL(k_flush_decide, dim3(1), dim3(1024), 0, s, d_ctr, ncounters, d_kminslot, d_weights, slots,
                       slot_begin, st, fb, enable, d_seal, seal_tag);
    return hipGetLastError();
}

hipError_t launch_tile_kmin(hipStream_t s, const float *d_k32, float *d_kmin32, int slots, int ntiles, size_t row_stride) {
    hipLaunchKernelGGL(k_tile_kmin, dim3(ntiles, slots), dim3(256), 0, s, d_k32, d_kmin32, ntiles, row_stride);
    return hipGetLastError();
}

hipError_t launch_cws_resolve(hipStream_t s, const double *d_rcb, const double *d_f64,
                              const float *d_tilemin, double *d_candA, int32_t *d_candB,
                              unsigned long long *d_mins, double *d_weights,
                              int slots, int slot_begin, int ntiles, const unsigned long long *d_scanmap, DevState *st,
                              const FlushBatch &fb) {
    const int merged = scan_merge_off() ? 0 : 1;                 // (this launcher is the no-drift path)
    if (merged) {
        prof_mark(s, "k_cws_resolve");
        hipLaunchKernelGGL(k_cws_resolve<true>, dim3(slots, 1), dim3(256), (size_t)ntiles * 4 * sizeof(float), s,
                           d_rcb, d_f64, d_tilemin, d_candA, d_candB, slots, ntiles, d_scanmap, (ntiles * 4 + 63) / 64, st, fb);
    }
    else {
        prof_mark(s, "k_cws_resolve");
        hipLaunchKernelGGL(k_cws_resolve<false>, dim3(slots, fb.count), dim3(256), (size_t)ntiles * 4 * sizeof(float), s,
                           d_rcb, d_f64, d_tilemin, d_candA, d_candB, slots, ntiles, d_scanmap, (ntiles * 4 + 63) / 64, st, fb);
    }
    prof_mark(s, "k_cws_apply");
    hipLaunchKernelGGL(k_cws_apply, dim3((slots + 255) / 256), dim3(256), 0, s, d_candA, d_candB, d_mins,
                       d_weights, slots, slot_begin, st, fb, merged);
    return hipGetLastError();
}

hipError_t launch_cws_resolve_drift(hipStream_t s, const double *d_rcb, const double *d_f64,
                                    const float *d_tilemin, unsigned long long *d_mins, double *d_weights,
                                    int slots, int slot_begin, int ntiles, double decay_weight, float *d_slotmin,
                                    const unsigned long long *d_scanmap, DevState *st, const FlushBatch &fb) {
    const int ngroups = (slots + SCAN_ROWS - 1) / SCAN_ROWS, wwords = (ntiles * 4 + 63) / 64;
    prof_mark(s, "k_slot_tmin");
    hipLaunchKernelGGL(k_slot_tmin, dim3(ngroups, fb.count), dim3(256), 0, s, d_tilemin, d_slotmin, ntiles * 4, ngroups,
                       d_scanmap, wwords, st, fb);
    prof_mark(s, "k_cws_resolve_drift");
    hipLaunchKernelGGL(k_cws_resolve_drift, dim3(slots), dim3(256), (size_t)ntiles * 4 * sizeof(float), s,
                       d_rcb, d_f64, d_tilemin, d_mins, d_weights, slots, slot_begin, ntiles, decay_weight, d_slotmin,
                       d_scanmap, wwords, st, fb);
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

hipError_t launch_alfg(hipStream_t s, const uint64_t *d_coef, const uint64_t *d_coef_far, uint32_t far_chunks,
                       uint64_t *d_windows, uint64_t *d_raw, uint32_t first_chunk, uint32_t n_chunks, uint64_t chunk_len) {
    // windows[first_chunk] is valid; produces windows[first_chunk + 1 .. first_chunk + n_chunks) and the chunks themselves
    if (d_coef_far && far_chunks > 1 && n_chunks > far_chunks) {
        hipLaunchKernelGGL(k_alfg_jump, dim3(1), dim3(640), 0, s, d_coef_far, d_windows, first_chunk, n_chunks, far_chunks,
                           0u, (n_chunks - 1) / far_chunks);
        hipLaunchKernelGGL(k_alfg_jump, dim3((n_chunks + far_chunks - 1) / far_chunks), dim3(640), 0, s, d_coef, d_windows,
                           first_chunk, n_chunks, 1u, far_chunks, far_chunks - 1);
    } else {
        hipLaunchKernelGGL(k_alfg_jump, dim3(1), dim3(640), 0, s, d_coef, d_windows, first_chunk, n_chunks, 1u, 0u, n_chunks);
    }
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

// sketches rounded up to the query tile: the slot-major arrays are [S][smash_padded_n(N)], zero rows behind N
uint32_t smash_padded_n(uint32_t N) { return (N + SMASH_TQ - 1) / SMASH_TQ * SMASH_TQ; }
hipError_t launch_smash(hipStream_t s, const unsigned long long *d_mins, const double *d_weights, uint32_t N, uint32_t S,
                        int metric, double *d_out, double *d_mT, double *d_wT) {
    if (N == 0) return hipSuccess;
    const uint32_t NP = smash_padded_n(N);
    hipLaunchKernelGGL(k_smash_prep, dim3((S + 31) / 32, NP / 32), dim3(256), 0, s, d_mins, d_weights, N, S, NP, d_mT, d_wT);
    const dim3 g(NP / SMASH_TQ, (N + SMASH_TS - 1) / SMASH_TS);
    if (metric == 1) hipLaunchKernelGGL(k_smash<1>, g, dim3(128), 0, s, d_mT, d_wT, N, NP, S, d_out);
    else hipLaunchKernelGGL(k_smash<0>, g, dim3(128), 0, s, d_mT, d_wT, N, NP, S, d_out);
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

}  // namespace hulk
