// hulk_flush.hip — orchestration of a batch behind the C ABI: the binning launches on the work stream, the two spectrum
// rings, the flush (count-min + CWS) on the flush stream, staging of host reads.  Reference seam: theBoss.AddSeq / Flush
// (src/pipeline/boss.go:24-36) under SeqMinimizer.Run's interval rule (src/pipeline/sketch.go:196-224).
#include "hulk_ctx.h"

#include <algorithm>
#include <thread>

namespace hulk {

namespace { thread_local hulk_ctx *g_prof_ctx = nullptr; }
ProfScope::ProfScope(hulk_ctx *c) : prev(g_prof_ctx) { g_prof_ctx = (c && (c->profiling & 16)) ? c : nullptr; }
ProfScope::~ProfScope() { g_prof_ctx = prev; }
void prof_mark(hipStream_t s, const char *kernel) {
    hulk_ctx *c = g_prof_ctx;
    if (!c) return;
    ProfMark m{nullptr, kernel, s};
    if (hipEventCreateWithFlags(&m.e, PROFILE_EVENT_FLAGS) != hipSuccess) return;
    if (hipEventRecord(m.e, s) != hipSuccess) { (void)hipEventDestroy(m.e); return; }
    c->marks.push_back(m);
}

hipError_t create_lane_stream(hipStream_t *s, int priority) {
    if (const char *e = HULK_EXP_ENV("HULK_K1_CU_FREE")) {       // experiment: the binning kernels may not use N of the CUs
        const int nfree = atoi(e);
        if (nfree > 0 && nfree < 256) {
            uint32_t mask[8];
            for (int i = 0; i < 8; i++) mask[i] = 0xffffffffu;
            const int every = 256 / nfree;
            for (int i = 0, n = 0; i < 256 && n < nfree; i += every, n++) mask[i >> 5] &= ~(1u << (i & 31));
            return hipExtStreamCreateWithCUMask(s, 8, mask);      // (default priority)
        }
    }
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, priority);
}

uint32_t *ring_hist(hulk_ctx *c) { return c->d_hist + (size_t)c->cur_ring * (size_t)c->ring_n * (size_t)c->B; }

// The work lane of spectrum ring r.  With two lanes (hulk_params.work_lanes, the default) every launch that touches ring r —
// binning, the deferred reads' generic kernel, the test hook — runs on lane r's stream, so consecutive batches, which
// alternate between the rings, alternate between two streams and are NOT ordered against each other: batch n+1's
// k_minimizer_fast (VALU + LDS, 4 waves per SIMD) runs beside batch n's k_jump_bin / k_jump_left / spectrum kernels (tails,
// low occupancy) and the two fill each other's issue bubbles — what two independent contexts fed alternately measured as
// +11 % (tools/archive/two_ctx_overlap.py).  Lane 0 is the context's stream, lane 1 a stream of its own (same priority).
hipStream_t lane_stream(hulk_ctx *c, int ring) {
    return (ring == 1 && c->lane[1].stream) ? c->lane[1].stream : c->stream;
}
static int lane_of_ring(const hulk_ctx *c, int ring) { return (c->work_lanes > 1 && !c->no_overlap) ? ring : 0; }
hipStream_t ring_stream(hulk_ctx *c) { return lane_stream(c, lane_of_ring(c, c->cur_ring)); }   // the lane of the current ring

// the work stream may only write spectra of the current ring once the flush that last read them is done
// a flush prepared on the current ring has to be queued before anything may wait for it (a partial interval keeps the
// next batch in the same ring)
int ring_issue_own_flush(hulk_ctx *c) {
    if (c->deferred.armed && c->deferred.ring == c->cur_ring) return issue_flush(c, nullptr);
    return HULK_OK;
}
// the event the work stream has to pass before it writes spectra of the current ring (null: nothing to wait for)
hipEvent_t ring_write_event(hulk_ctx *c) {
    if (!c->pending_flush[c->cur_ring]) return nullptr;
    c->pending_flush[c->cur_ring] = false;
    return c->ev_flushed[c->cur_ring];
}
int ring_ready_for_writes(hulk_ctx *c) {
    { const int rc = ring_issue_own_flush(c); if (rc != HULK_OK) return rc; }
    if (hipEvent_t e = ring_write_event(c)) HIPCHK(c, hipStreamWaitEvent(ring_stream(c), e, 0));
    return HULK_OK;
}

// HULK_ERR_HIP / HULK_ERR_COMM from a step are fatal (hulk_hip.h): a stream of this context may be waiting for a collective
// the peers have left, so nothing is queued or waited for any more — every entry point returns the status
int fatal_status(hulk_ctx *c) {
    if (c->sticky == HULK_ERR_HIP || c->sticky == HULK_ERR_COMM) return fail(c, c->sticky);
    return HULK_OK;
}

int sync_all(hulk_ctx *c) {
    { const int rc = fatal_status(c); if (rc != HULK_OK) return rc; }
    { const int rc = issue_flush(c, nullptr); if (rc != HULK_OK) return rc; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->lane[1].stream) HIPCHK(c, hipStreamSynchronize(c->lane[1].stream));
    c->stagger = 1;                                             // both lanes idle: see bin_fast
    HIPCHK(c, hipStreamSynchronize(c->flush_stream));
    return HULK_OK;
}

// kernel configuration by read length: {xcap, table, block threads}
// the one-wave-per-read kernel takes reads of up to 1024 k-mer positions; its 4096-position configuration
// ran at 10 Gbases/s (32 KB of LDS per wave), the grouped long-sequence path does 21 — so longer reads go there
constexpr uint32_t GENERIC_XCAP_MAX = 1024;
// returns false when some reads may exceed the largest configuration (they take the long-read path)
bool pick_config(uint32_t k, uint32_t max_len, MinimizerParams &P, int &threads) {
    const uint32_t npos = max_len >= k ? max_len - k + 1 : 1;
    if (npos <= 192) { P.xcap = 192; P.tab_size = 256; threads = 256; return true; }
    P.xcap = GENERIC_XCAP_MAX; P.tab_size = 2048; threads = 64;
    return npos <= GENERIC_XCAP_MAX;
}

// sequences with more than GENERIC_XCAP_MAX k-mer positions: grouped launches of the long-sequence kernels
int bin_long_reads(hulk_ctx *c, hipStream_t s, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n, MinimizerParams P,
                   uint32_t *hist, const uint64_t *h_offsets) {
    // the lengths: from the caller's host copy of the offsets when there is one (ctx_hint_host_offsets), else fetched from the
    // device — which waits for everything queued on the lane before
    std::vector<uint64_t> fetched;
    const uint64_t *off = h_offsets;
    if (!off) {
        fetched.resize(n + 1);
        HIPCHK(c, hipMemcpyAsync(fetched.data(), d_offsets, (n + 1) * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        off = fetched.data();
    }
    // groups of long sequences, one launch set per group: bounded scratch (positions) and grid.y
    constexpr uint64_t GROUP_POS = 128ull << 20;        // positions per group (8 B + 1 B scratch, <= 16 B of table each)
    constexpr uint32_t GROUP_SEQS = 32768;
    std::vector<hulk::LongSeqDesc> descs;
    uint64_t pos_total = 0, tab_total = 0, max_npos = 0;
    auto launch_group = [&]() -> int {
        if (descs.empty()) return HULK_OK;
        static const bool two_pass = HULK_EXP_ENV("HULK_LONG_TWO_PASS") != nullptr;      // the round-1..5 form as comparator (9 B of scratch per position)
        if (two_pass && pos_total > c->long_cap) {
            HIPCHK(c, hipStreamSynchronize(s));
            if (c->long_pending) HIPCHK(c, hipEventSynchronize(c->ev_long));
            hipFree(c->d_long_xs); hipFree(c->d_long_valid); c->d_long_xs = nullptr; c->d_long_valid = nullptr; c->long_cap = 0;
            HIPCHK(c, hipMalloc((void **)&c->d_long_xs, pos_total * 8));
            HIPCHK(c, hipMalloc((void **)&c->d_long_valid, pos_total));
            c->long_cap = pos_total;
        }
        if (tab_total > c->long_table_cap) {
            HIPCHK(c, hipStreamSynchronize(s));
            if (c->long_pending) HIPCHK(c, hipEventSynchronize(c->ev_long));
            hipFree(c->d_long_table); c->d_long_table = nullptr; c->long_table_cap = 0;
            HIPCHK(c, hipMalloc((void **)&c->d_long_table, tab_total * 8));
            c->long_table_cap = tab_total;
        }
        if (descs.size() > c->long_desc_cap) {
            HIPCHK(c, hipStreamSynchronize(s));
            if (c->long_pending) HIPCHK(c, hipEventSynchronize(c->ev_long));
            hipFree(c->d_long_desc); c->d_long_desc = nullptr; c->long_desc_cap = 0;
            HIPCHK(c, hipMalloc((void **)&c->d_long_desc, (descs.size() + 1024) * sizeof(hulk::LongSeqDesc)));
            c->long_desc_cap = descs.size() + 1024;
        }
        if (c->long_pending && c->long_last_stream != s && !HULK_EXP_ENV("HULK_LONG_NO_LANE_ORDER"))          // (the switch: to show that the test for it fails)
            HIPCHK(c, hipStreamWaitEvent(s, c->ev_long, 0));                                                   // the other lane's group: same scratch
        // through one of two pinned staging buffers (the copy runs when the lane gets there: the source has to stay as it is)
        hulk_ctx::LongDescStage &H = c->h_long_desc[c->long_desc_turn];
        c->long_desc_turn ^= 1;
        if (H.used) HIPCHK(c, hipEventSynchronize(H.ev));          // the copy out of it, two groups ago
        const size_t dbytes = descs.size() * sizeof(hulk::LongSeqDesc);
        if (dbytes > H.cap) {
            if (H.p) HIPCHK(c, hipHostFree(H.p));
            H.p = nullptr; H.cap = 0;
            HIPCHK(c, hipHostMalloc(&H.p, dbytes + 1024 * sizeof(hulk::LongSeqDesc), hipHostMallocDefault));
            H.cap = dbytes + 1024 * sizeof(hulk::LongSeqDesc);
        }
        if (!H.ev) HIPCHK(c, hipEventCreateWithFlags(&H.ev, hipEventDisableTiming));
        memcpy(H.p, descs.data(), dbytes);
        HIPCHK(c, hipMemcpyAsync(c->d_long_desc, H.p, dbytes, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipEventRecord(H.ev, s));
        H.used = true;
        HIPCHK(c, launch_long_group(s, d_bases, (const hulk::LongSeqDesc *)c->d_long_desc, (uint32_t)descs.size(),
                                    max_npos, P, c->d_long_xs, c->d_long_valid, c->d_long_table, tab_total, hist,
                                    c->d_min_slots));
        if (!c->ev_long) HIPCHK(c, hipEventCreateWithFlags(&c->ev_long, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(c->ev_long, s));
        c->long_last_stream = s; c->long_pending = true;
        descs.clear(); pos_total = tab_total = max_npos = 0;
        return HULK_OK;
    };
    for (uint64_t rd = 0; rd < n; rd++) {
        const uint64_t L = off[rd + 1] - off[rd];
        if (L < (uint64_t)P.k || L - P.k + 1 <= GENERIC_XCAP_MAX) continue;
        const uint64_t npos = L - P.k + 1;
        uint64_t tsize = 1; while (tsize < npos) tsize <<= 1;      // <= ~0.2 distinct minimizers per position: load <= 0.2
        if (!descs.empty() && (pos_total + npos > GROUP_POS || descs.size() >= GROUP_SEQS)) {
            const int rc = launch_group();
            if (rc != HULK_OK) return rc;
        }
        hulk::LongSeqDesc d{};
        d.seq_off = off[rd]; d.L = L; d.xs_off = pos_total; d.tab_off = tab_total; d.tab_mask = tsize - 1;
        d.hslot = P.ring_base;
        if (P.interval) d.hslot = (uint32_t)(((P.fill + rd) / P.interval + P.ring_base) % P.ring_n);
        descs.push_back(d);
        pos_total += npos; tab_total += tsize; if (npos > max_npos) max_npos = npos;
    }
    return launch_group();
}

// grow-only buffers of a work lane for a launch of n reads
static int lane_reserve(hulk_ctx *c, hulk_ctx::BinLane &ln, hipStream_t s, uint64_t n, bool pair_ok) {
    if (!ln.d_slow_count) {
        HIPCHK(c, dalloc(&ln.d_slow_count, 2));
        HIPCHK(c, hipMemsetAsync(ln.d_slow_count, 0, 8, s));
    }
    if (n > ln.d_slow_cap) {
        HIPCHK(c, hipStreamSynchronize(s));
        hipFree(ln.d_slow_list); ln.d_slow_list = nullptr; ln.d_slow_cap = 0;
        HIPCHK(c, hipMalloc((void **)&ln.d_slow_list, (size_t)(n + n / 4 + 1024) * 4));
        ln.d_slow_cap = n + n / 4 + 1024;
    }
    MinimizerList &ml = ln.ml;
    const uint64_t regions = (n + FAST_READS_PER_WAVE - 1) / FAST_READS_PER_WAVE;
    // (a region never shrinks again: calls with and without reads of two groups may alternate)
    const uint64_t rcap = std::max<uint64_t>(minimizer_list_rcap(c->p.w, pair_ok), ml.rcap);
    if (regions > ln.ml_regions || ml.rcap != rcap) {
        HIPCHK(c, hipStreamSynchronize(s));
        hipFree(ml.x); hipFree(ml.slot); hipFree(ml.key); hipFree(ml.cnt); hipFree(ml.off); hipFree(ml.bsum);
        hipFree(ml.lo); hipFree(ml.lo_cnt); hipFree(ml.dmask); hipFree(ml.dsum);
        uint32_t *keep_partial = ml.partial; const uint32_t keep_parts = ml.max_parts;
        uint32_t *keep_nib = ml.nib, *keep_over = ml.nib_over; const uint32_t keep_np = ml.nib_parts;
        ml = MinimizerList{}; ln.ml_regions = 0;
        ml.partial = keep_partial; ml.max_parts = keep_parts;
        ml.nib = keep_nib; ml.nib_over = keep_over; ml.nib_parts = keep_np;
        const uint64_t cap = regions + regions / 8 + 64;
        HIPCHK(c, hipMalloc((void **)&ml.x, cap * rcap * 8));
        HIPCHK(c, hipMalloc((void **)&ml.slot, cap * rcap));
        HIPCHK(c, hipMalloc((void **)&ml.key, cap * rcap * 4));
        HIPCHK(c, hipMalloc((void **)&ml.cnt, cap * 4));
        HIPCHK(c, hipMalloc((void **)&ml.off, (cap + 1) * 4));
        HIPCHK(c, hipMalloc((void **)&ml.bsum, (cap / 1024 + 2) * 4));
        HIPCHK(c, hipMalloc((void **)&ml.lo, cap * JUMP_LO_CAP * sizeof(uint4)));
        HIPCHK(c, hipMalloc((void **)&ml.lo_cnt, cap * 4));
        HIPCHK(c, hipMalloc((void **)&ml.dmask, cap * 4));
        HIPCHK(c, hipMalloc((void **)&ml.dsum, (cap / 1024 + 2) * 4));
        if (!ml.nib) {
            const size_t nr = ((size_t)c->B + 262143) / 262144;
            ml.nib_parts = 48;
            HIPCHK(c, hipMalloc((void **)&ml.nib, (size_t)ml.nib_parts * c->ring_n * nr * (262144 / 8) * 4));
            HIPCHK(c, hipMalloc((void **)&ml.nib_over, RING_MAX * 4));
            HIPCHK(c, hipMemsetAsync(ml.nib_over, 0, RING_MAX * 4, s));
        }
        if (!ml.partial) {
            ml.max_parts = 8;
            HIPCHK(c, hipMalloc((void **)&ml.partial, (size_t)ml.max_parts * c->ring_n * (size_t)c->B * 4));
        }
        ml.rcap = rcap; ln.ml_regions = cap;
    }
    return HULK_OK;
}

// A launch chain of the short-read kernels on stream s with lane ln's buffers: k_minimizer_fast -> region scan ->
// k_jump_bin / k_jump_left -> spectrum kernels -> the generic kernel over the reads the fast one deferred.
// spectra_gate (may be null): the flush that last read this ring — only the histogram kernels wait for it (the minimizer
// and jump-hash kernels do not touch the spectra)
static int bin_fast(hulk_ctx *c, hulk_ctx::BinLane &ln, hipStream_t s, const uint8_t *d_bases, const uint64_t *d_offsets,
                    uint64_t n, uint32_t max_len, MinimizerParams P, uint32_t *hist, hipEvent_t spectra_gate) {
    { const int rc = lane_reserve(c, ln, s, n, P.pair != 0); if (rc != HULK_OK) return rc; }
    ProfScope prof_scope(c);
    // Both lanes idle (the context was just synchronised): the two batches that come next would start their
    // k_minimizer_fast together and run in lock step — same kernels side by side, nothing to fill — for several batches
    // before they drift apart.  The second one therefore waits for the first one's k_minimizer_fast, which puts the lanes
    // half a batch apart at once (k_minimizer_fast beside the other lane's k_jump_bin).
    if (c->stagger == 2 && c->ev_stagger) { HIPCHK(c, hipStreamWaitEvent(s, c->ev_stagger, 0)); c->stagger = 0; }
    ProfileRec pr{}; pr.which = 1;
    if ((c->profiling & 2)) {
        HIPCHK(c, hipEventCreateWithFlags(&pr.a, PROFILE_EVENT_FLAGS)); HIPCHK(c, hipEventCreateWithFlags(&pr.b, PROFILE_EVENT_FLAGS));
        HIPCHK(c, hipEventRecord(pr.a, s));
    }
    if (c->scaling && HULK_EXP_ENV("HULK_C3_GATE")) {
        // experiment: k_minimizer_fast does not start while the 112 KB-LDS kernels of the flush queued TWO batches ago (the one that
        // ran beside the previous batch's jump-hash kernels) still hold the CUs' LDS — it would run at a quarter of its occupancy
        const int older = c->heavy_idx ^ 1;
        if (c->heavy_set[older]) HIPCHK(c, hipStreamWaitEvent(s, c->ev_heavy[older], 0));
    }
    HIPCHK(c, launch_minimizer_fast(s, d_bases, d_offsets, n, P, ln.ml, c->d_state, c->d_min_slots));
    if ((c->profiling & 2)) { HIPCHK(c, hipEventRecord(pr.b, s)); c->prof.push_back(pr); }
    if (c->deferred.armed && c->scaling && HULK_EXP_ENV("HULK_C3_HOLD")) {
        // experiment: the flush of the batch before was held back (flush_batch) and is queued now, behind THIS batch's k_minimizer_fast:
        // its 112 KB-LDS replay kernels then run beside k_jump_bin, which needs no LDS, and k_minimizer_fast keeps its occupancy
        if (!c->ev_hold) HIPCHK(c, hipEventCreateWithFlags(&c->ev_hold, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(c->ev_hold, s));
        const int rch = issue_flush(c, c->ev_hold);
        if (rch != HULK_OK) return rch;
    }
    if (c->stagger == 1 && c->work_lanes > 1 && !c->no_overlap) {
        if (!c->ev_stagger) HIPCHK(c, hipEventCreateWithFlags(&c->ev_stagger, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(c->ev_stagger, s));
        c->stagger = 2;
    }
    ProfileRec pj{}; pj.which = 2;
    ProfileRec pl{}; pl.which = 3;
    if ((c->profiling & 4)) {
        HIPCHK(c, hipEventCreateWithFlags(&pj.a, PROFILE_EVENT_FLAGS)); HIPCHK(c, hipEventCreateWithFlags(&pj.b, PROFILE_EVENT_FLAGS));
        HIPCHK(c, hipEventCreateWithFlags(&pl.a, PROFILE_EVENT_FLAGS)); HIPCHK(c, hipEventCreateWithFlags(&pl.b, PROFILE_EVENT_FLAGS));
    }
    HIPCHK(c, launch_minimizer_post(s, n, P, ln.ml, hist, ln.d_slow_list, ln.d_slow_count, pj.a, pj.b, spectra_gate, pl.a, pl.b));
    if ((c->profiling & 4)) { c->prof.push_back(pj); c->prof.push_back(pl); }
    int threads = 256;
    pick_config(c->p.k, max_len, P, threads);      // (the fast path implies max_len <= 512: always fits)
    // the list is normally empty or short; its length is only known on the device, so the grid is fixed: enough
    // workgroups that 1 % of deferred reads (reads with N) do not queue behind 512 waves (blocks past the list exit at once)
    static const uint32_t slow_blocks = [] { const char *e = HULK_EXP_ENV("HULK_SLOW_BLOCKS"); const long v = e ? atol(e) : 2048; return (uint32_t)(v < 1 ? 1 : v > 8192 ? 8192 : v); }();
    const uint32_t list_blocks = (uint32_t)std::min<uint64_t>(slow_blocks, (n + 3) / 4);
    HIPCHK(c, launch_minimizer_bin(s, d_bases, d_offsets, n, P, threads, hist, c->d_state, c->d_min_slots, ln.d_slow_list,
                                   ln.d_slow_count, list_blocks));
    prof_mark(s, "-");
    return HULK_OK;
}

// launch parameters of a call that starts `fill` reads into spectrum ring_base
static MinimizerParams piece_params(const hulk_ctx *c, uint64_t bases_bytes, uint64_t interval, uint64_t fill, bool pair) {
    MinimizerParams P{};
    P.k = c->p.k; P.w = c->p.w; P.num_bins = c->B; P.bases_bytes = bases_bytes;
    P.interval = interval; P.fill = fill; P.ring_base = c->ring_base; P.ring_n = c->ring_n;
    // whole intervals in front of this launch move the first spectrum, not the fill: the kernels then see a launch that
    // starts inside spectrum ring_base (hist_slot() is unchanged by this) and build no empty spectra in front of it
    if (P.interval && P.fill >= P.interval) { P.ring_base = (uint32_t)((P.ring_base + P.fill / P.interval) % P.ring_n); P.fill %= P.interval; }
    if (const char *e = HULK_EXP_ENV("HULK_K1_DEBUG")) P.debug = (uint32_t)atoi(e);
    P.pair = pair ? 1u : 0u;
    return P;
}

// lane 1 comes into being with the first batch of ring 1
static int lane_open(hulk_ctx *c, int li) {
    if (li == 0 || c->lane[1].stream) return HULK_OK;
    int prio = 0;
    if (c->stream) (void)hipStreamGetPriority(c->stream, &prio);
    HIPCHK(c, create_lane_stream(&c->lane[1].stream, prio));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    return HULK_OK;
}
// hulk_create: the work lanes' buffers (minimizer list, hand-over areas, 4-bit histogram parts: ~3 GB per lane at the
// defaults) sized for the largest batch the interval rule hands to bin_reads, so that no call of the stream allocates.
// (hipMalloc in the middle of a stream was measured at anything between 0.1 ms and 1.8 SECONDS, depending on what the
// process freed before: profiles/r04_bench_ramp.txt.)  Without an interval a call may be of any size: the first one sizes
// the lists, as any later call that needs more does.
int lanes_prereserve(hulk_ctx *c) {
    const uint64_t I = c->p.interval;
    if (!I || (c->p.flags & HULK_FLAG_NO_PRERESERVE) || HULK_EXP_ENV("HULK_NO_PRERESERVE")) return HULK_OK;
    const uint64_t n = std::min<uint64_t>((uint64_t)c->T * I, MAX_READS_PER_LAUNCH);
    const int nl = (c->work_lanes > 1 && !c->no_overlap) ? 2 : 1;
    for (int li = 0; li < nl; li++) {
        { const int rc = lane_open(c, li); if (rc != HULK_OK) return rc; }
        const int rc = lane_reserve(c, c->lane[li], li ? c->lane[1].stream : c->stream, n, false);
        if (rc != HULK_OK) return rc;
    }
    return HULK_OK;
}
// the context's stream has passed everything lane 1 was given so far
int lanes_join(hulk_ctx *c) {
    if (!c->lane[1].stream) return HULK_OK;
    HIPCHK(c, hipEventRecord(c->ev_join, c->lane[1].stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    return HULK_OK;
}

// Bins reads [0, n) of a call into the spectra of the current ring (read i -> spectrum (fill + i) / interval), on the
// ring's work lane.  What the caller queued on the context's stream so far (its buffers, the host path's copies) is
// passed to lane 1 through an event.  `join`: the context's stream waits for the lane before the call returns — always
// when the context runs on a caller's stream (hulk_set_stream: to the caller it stays ONE stream, at the price of the
// overlap between batches), and for the entry points whose results the caller reads on that stream (hulk_bin_reads_device).
int bin_reads(hulk_ctx *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n,
              uint32_t max_len, uint64_t bases_bytes, uint64_t interval, uint64_t fill, bool join) {
    { int rcw = ring_issue_own_flush(c); if (rcw != HULK_OK) return rcw; }
    const int li = lane_of_ring(c, c->cur_ring);
    { const int rc = lane_open(c, li); if (rc != HULK_OK) return rc; }
    hipStream_t s = li ? c->lane[1].stream : c->stream;
    // what was queued on the context's stream for this batch — the caller's own work when the stream is the caller's, the
    // host path's copies into the staging set (copies_pending) — has to be passed to lane 1.  Nothing else is: a fork in
    // front of every batch would order lane 1 behind the batch lane 0 was just given, which is the overlap itself
    if (li && (c->stream != c->own_stream || c->copies_pending)) {
        HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
        HIPCHK(c, hipStreamWaitEvent(s, c->ev_fork, 0));
        c->copies_pending = false;
    }
    c->last_bin_stream = s;
    uint32_t *hist = ring_hist(c);
    // the short-read kernel takes reads of <= 16*w k-mer positions and <= 256 bases; when the batch's
    // length bound already exceeds that, go straight to the generic kernel
    // ... or, two groups per read, <= 2*16w - (w-1) positions (300 bases at k = 21, w = 9) while a group's own
    // 16w + k - 1 bases fit its 256-base staging
    const bool fast_base = c->p.w >= 1 && c->p.w <= 16 && !HULK_EXP_ENV("HULK_NO_FAST_K1") && n < 0xffffffffull;
    const bool single_ok = max_len <= 256 && (uint64_t)max_len < (uint64_t)c->p.k + 16ull * c->p.w;
    const bool pair_ok = !single_ok && !HULK_EXP_ENV("HULK_NO_PAIR") && 16ull * c->p.w + c->p.k - 1 <= 256 && max_len <= 512 &&
                         (uint64_t)max_len < (uint64_t)c->p.k + 32ull * c->p.w - (c->p.w - 1);
    int rc = HULK_OK;
    if (fast_base && (single_ok || pair_ok)) {
        // short-read kernel first; reads it cannot take (N bases, too long for 16 blocks of w
        // positions) are queued on the device and binned by the generic kernel right after
        rc = bin_fast(c, c->lane[li], s, d_bases, d_offsets, n, max_len, piece_params(c, bases_bytes, interval, fill, pair_ok),
                      hist, ring_write_event(c));
    } else {
        MinimizerParams P = piece_params(c, bases_bytes, interval, fill, false);
        int threads = 256;
        ProfScope prof_scope(c);
        struct Close { hipStream_t s; ~Close() { prof_mark(s, "-"); } } prof_close{s};
        rc = issue_flush(c, nullptr);
        if (rc == HULK_OK) {
            if (hipEvent_t e = ring_write_event(c)) HIPCHK(c, hipStreamWaitEvent(s, e, 0));
            const bool fits = pick_config(c->p.k, max_len, P, threads);
            P.skip_long = fits ? 0u : 1u;
            HIPCHK(c, launch_minimizer_bin(s, d_bases, d_offsets, n, P, threads, hist, c->d_state,
                                           c->d_min_slots, nullptr, nullptr, 0));
            if (!fits) rc = bin_long_reads(c, s, d_bases, d_offsets, n, P, hist, c->h_off_chunk);
        }
    }
    if (rc != HULK_OK) return rc;
    if (li && (join || c->stream != c->own_stream)) return lanes_join(c);
    return HULK_OK;
}

// The kernels of one flush: `fb.count` consecutive spectra of `hist` (starting at fb.ring_base) through count-min + CWS, on stream s.
int flush_kernels(hulk_ctx *c, hipStream_t s, uint32_t *hist, const FlushBatch &fb) {
    ProfScope prof_scope(c);
    struct Close { hipStream_t s; ~Close() { prof_mark(s, "-"); } } prof_close{s};
    if (!c->scaling) HIPCHK(c, launch_count_used(s, hist, c->d_state, fb));    // (with decay k_elem_index delivers the count)
    {   // whole-batch bound on the counters as they stand BEFORE this batch is added (see k_flush_decide)
        HIPCHK(c, launch_flush_decide(s, c->d_ctr, c->cms_depth * c->cms_width, c->d_kminslot, c->d_weights, (int)c->slots,
                                      (int)c->slot_begin, c->d_state, fb, (c->prune && !c->drift && !c->no_skip && c->slots) ? 1 : 0));
    }
    if (c->scaling) {
        HIPCHK(c, launch_elem_index(s, hist, c->d_blkcnt, c->d_eidx, c->d_etot, fb, c->d_state));
        ProfileRec pf{}; pf.which = 4;
        if ((c->profiling & 8)) { HIPCHK(c, hipEventCreateWithFlags(&pf.a, PROFILE_EVENT_FLAGS)); HIPCHK(c, hipEventCreateWithFlags(&pf.b, PROFILE_EVENT_FLAGS)); }
        HIPCHK(c, launch_cmsd_binorder(s, hist, c->d_pos16, c->d_meta8, c->d_eidx, c->d_etot, c->d_ctrd, c->d_segadd,
                                       c->d_segfac, c->d_sege0, c->d_cstart, c->d_f64, c->d_rcp32, c->cms_depth,
                                       c->cms_width, c->row_stride, c->decay_weight, c->d_state, fb, pf.a, pf.b, c->cms_chain));
        if ((c->profiling & 8)) c->prof.push_back(pf);
        if (HULK_EXP_ENV("HULK_C3_GATE")) {
            c->heavy_idx ^= 1;
            if (!c->ev_heavy[c->heavy_idx]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_heavy[c->heavy_idx], hipEventDisableTiming));
            HIPCHK(c, hipEventRecord(c->ev_heavy[c->heavy_idx], s));
            c->heavy_set[c->heavy_idx] = true;
        }
    } else {
        HIPCHK(c, launch_cms_binorder(s, hist, c->d_pos16, c->d_meta8, c->d_ctr, c->d_segsum, c->d_cbase, c->d_f64,
                                      c->d_rcp32, c->cms_depth, c->cms_width, c->row_stride, c->d_state, fb, c->cms_chain));
    }
    if (c->slots) {
        ProfileRec pr{};
        if ((c->profiling & 1)) { HIPCHK(c, hipEventCreateWithFlags(&pr.a, PROFILE_EVENT_FLAGS)); HIPCHK(c, hipEventCreateWithFlags(&pr.b, PROFILE_EVENT_FLAGS)); }
        HIPCHK(c, launch_cws_scan(s, c->d_k32, c->d_rcp32, c->d_tilemin, (int)c->slots, c->ntiles,
                                  c->row_stride, c->d_state, fb, c->prune ? c->d_kmin32 : nullptr, c->d_rext,
                                  c->d_weights, (int)c->slot_begin, c->d_visited, c->drift ? c->decay_weight : 0.0, c->d_scanmap,
                                  c->drift /* per-interval minima: the drift resolve replays the stream in order */, c->d_rmm,
                                  c->d_scanlist, c->d_scanlist_n, pr.a, pr.b));     // (the brackets: around the scan kernel itself)
        c->scan_tiles_total += (uint64_t)((c->slots + SCAN_ROWS - 1) / SCAN_ROWS) * (uint64_t)c->ntiles * 4u;
        if ((c->profiling & 1)) c->prof.push_back(pr);
        if (c->drift)
            HIPCHK(c, launch_cws_resolve_drift(s, c->d_rcb, c->d_f64, c->d_tilemin, c->d_mins, c->d_weights, (int)c->slots,
                                               (int)c->slot_begin, c->ntiles, c->decay_weight, c->d_slotmin, c->d_scanmap, c->d_state, fb));
        else
        HIPCHK(c, launch_cws_resolve(s, c->d_rcb, c->d_f64, c->d_tilemin, c->d_candA, c->d_candB, c->d_mins, c->d_weights,
                                     (int)c->slots, (int)c->slot_begin, c->ntiles, c->d_scanmap, c->d_state, fb));
    }
    return HULK_OK;
}

// the stream flushes run on (HULK_NO_OVERLAP: the work stream itself — profiling aid, every kernel alone)
bool no_overlap_mode(const hulk_ctx *c) { return c->no_overlap; }
hipStream_t flush_stream_of(hulk_ctx *c) { return c->no_overlap ? c->stream : c->flush_stream; }

// queue the kernels of a prepared flush on the flush stream; `gate` (may be null): an event on the work stream they wait for
int issue_flush(hulk_ctx *c, hipEvent_t gate) {
    if (!c->deferred.armed) return HULK_OK;
    c->deferred.armed = false;
    const FlushBatch fb = c->deferred.fb;
    const int ring = c->deferred.ring;
    hipStream_t s = flush_stream_of(c);
    // ev_binned orders the flush behind the binning; recorded on a caller's stream (hulk_flush_batch_after: the stream its
    // collective runs on) it has to be waited for even when the flush shares the work stream
    if (!no_overlap_mode(c) || c->deferred.use_dep) HIPCHK(c, hipStreamWaitEvent(s, c->ev_binned, 0));
    if (!no_overlap_mode(c) && gate) HIPCHK(c, hipStreamWaitEvent(s, gate, 0));
    uint32_t *hist = c->d_hist + (size_t)ring * (size_t)c->ring_n * (size_t)c->B;
    if (c->deferred.allreduce) {                                 // hulk_step_sliced: sum the ranks' spectra first
        const int rc = comm_allreduce_u32(c, s, hist + (size_t)fb.ring_base * (size_t)c->B, (size_t)fb.count * (size_t)c->B);
        if (rc != HULK_OK) return rc;
    }
    { const int rc = flush_kernels(c, s, hist, fb); if (rc != HULK_OK) return rc; }
    HIPCHK(c, hipEventRecord(c->ev_flushed[ring], s));
    c->pending_flush[ring] = true;
    return HULK_OK;
}

// Flush `count` consecutive spectra of the ring (starting at ring_base) through count-min + CWS.
int flush_batch(hulk_ctx *c, uint32_t count, hipStream_t dep_stream, bool use_dep, bool allreduce) {
    { const int rcf = fatal_status(c); if (rcf != HULK_OK) return rcf; }
    if (count == 0) return HULK_OK;
    int rc = ensure_tables(c);
    if (rc != HULK_OK) return rc;
    rc = issue_flush(c);                                        // (at most one flush is ever waiting)
    if (rc != HULK_OK) return rc;
    FlushBatch fb{};
    fb.ring_base = c->ring_base; fb.ring_n = c->ring_n; fb.count = count;
    fb.parity = (int)(c->flush_index & 1); fb.num_bins = c->B;
    // everything binned so far (or the caller's all-reduce on dep_stream) ends where this event is recorded
    HIPCHK(c, hipEventRecord(c->ev_binned, use_dep ? dep_stream : ring_stream(c)));
    c->deferred.armed = true; c->deferred.fb = fb; c->deferred.ring = c->cur_ring;
    c->deferred.use_dep = use_dep; c->deferred.allreduce = allreduce;
    c->flush_index++;
    // Queued at once.  (Holding the flush back until the NEXT batch's minimizer kernel had run — so that its LDS-heavy
    // count-min kernels would meet k_jump_bin, which needs no LDS, instead of k_minimizer_fast — was measured: C3-shaped
    // 8.9e8 vs 9.8e8 reads/s without the delay.  What does pay is that the next batch's minimizer and jump-hash kernels
    // no longer wait for this flush: only the histogram kernels behind them do, see bin_reads.)
    if (c->scaling && !c->no_overlap && HULK_EXP_ENV("HULK_C3_HOLD")) return HULK_OK;       // (experiment: see bin_fast)
    return issue_flush(c);
}


int check_device_error(hulk_ctx *c) {
    { int rcs = sync_all(c); if (rcs != HULK_OK) return rcs; }
    DevState st{};
    HIPCHK(c, hipMemcpyAsync(&st, c->d_state, sizeof st, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (st.err != 0) { c->sticky = st.err; return fail(c, st.err); }
    return HULK_OK;
}


// internal accessors for hulk_ingest.hip (not part of the ABI)
hipStream_t ctx_stream(hulk_ctx *c) { return c->stream; }
uint64_t ctx_min_read_len(const hulk_ctx *c) { return (uint64_t)c->p.w + c->p.k - 1; }
int ctx_fail(hulk_ctx *c, int code, const char *full_message) {
    c->last_error = (full_message && *full_message) ? full_message : err_text(code);
    return code;
}

// a staging set that is free again (its last copies and kernels done) and holds nbytes of bases and cn reads
int stage_ready(hulk_ctx *c, hulk_ctx::HostStage &hs, size_t nbytes, uint64_t cn) {
    if (!hs.ev) HIPCHK(c, hipEventCreateWithFlags(&hs.ev, hipEventDisableTiming));
    if (!hs.ev1) HIPCHK(c, hipEventCreateWithFlags(&hs.ev1, hipEventDisableTiming));
    if (hs.busy) { HIPCHK(c, hipEventSynchronize(hs.ev)); hs.busy = false; }     // its copies and kernels are done
    if (hs.busy1) { HIPCHK(c, hipEventSynchronize(hs.ev1)); hs.busy1 = false; }  // ... on both work lanes
    if (nbytes + 32 > hs.cap_bases) {
        if (hs.h_bases) hipHostFree(hs.h_bases);
        hipFree(hs.d_bases); hs.h_bases = hs.d_bases = nullptr;
        hs.cap_bases = (nbytes + 32) + (nbytes + 32) / 4;
        HIPCHK(c, hipHostMalloc((void **)&hs.h_bases, hs.cap_bases, hipHostMallocDefault));
        HIPCHK(c, hipMalloc((void **)&hs.d_bases, hs.cap_bases));
    }
    if (cn + 2 > hs.cap_off) {
        if (hs.h_off) hipHostFree(hs.h_off);
        hipFree(hs.d_off); hs.h_off = hs.d_off = nullptr;
        hs.cap_off = (cn + 2) + (cn + 2) / 4;
        HIPCHK(c, hipHostMalloc((void **)&hs.h_off, hs.cap_off * 8, hipHostMallocDefault));
        HIPCHK(c, hipMalloc((void **)&hs.d_off, hs.cap_off * 8));
    }
    return HULK_OK;
}
// the copies (context's stream) and the kernels (either work lane) that read the set have been queued: it is free again when
// both streams have passed this point
int stage_mark_busy(hulk_ctx *c, hulk_ctx::HostStage &hs) {
    HIPCHK(c, hipEventRecord(hs.ev, c->stream));
    hs.busy = true;
    if (c->lane[1].stream) { HIPCHK(c, hipEventRecord(hs.ev1, c->lane[1].stream)); hs.busy1 = true; }
    return HULK_OK;
}
// reads [i0, i1) of the caller's host buffers -> the next of the two pinned + device staging sets: host copy (several
// threads: one core copies ~10 GB/s, a PCIe 5 x16 link moves ~50) and hipMemcpyAsync on the context's stream.  The caller
// queues its kernels behind the copies, then records hs.ev and sets hs.busy (the set is reused when that event has passed).
int stage_host_reads(hulk_ctx *c, const uint8_t *bases, const uint64_t *offsets, uint64_t i0, uint64_t i1,
                     hulk_ctx::HostStage **out) {
    const uint64_t cn = i1 - i0, lo = offsets[i0];
    const size_t nbytes = (size_t)(offsets[i1] - lo);
    hulk_ctx::HostStage &hs = c->hstage[c->hstage_cur];
    { const int rc = stage_ready(c, hs, nbytes, cn); if (rc != HULK_OK) return rc; }
    {
        const unsigned T = nbytes >= (8u << 20) ? c->host_copy_threads : 1u;
        // (ceil: with floor(nbytes / T) a multiple of 64 and nbytes % T != 0 the T pieces would end short of the last bytes)
        const size_t piece = ((nbytes + T - 1) / T + 63) & ~(size_t)63;
        std::vector<std::thread> th;
        auto work = [&](unsigned t) {
            const size_t at = (size_t)t * piece;
            if (at < nbytes) memcpy(hs.h_bases + at, bases + lo + at, std::min(piece, nbytes - at));
        };
        for (unsigned t = 1; t < T; t++) th.emplace_back(work, t);
        work(0);
        for (uint64_t i = 0; i <= cn; i++) hs.h_off[i] = offsets[i0 + i] - lo;
        for (auto &x : th) x.join();
    }
    c->copies_pending = true;
    HIPCHK(c, hipMemcpyAsync(hs.d_bases, hs.h_bases, nbytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(hs.d_off, hs.h_off, (cn + 1) * 8, hipMemcpyHostToDevice, c->stream));
    c->hstage_cur ^= 1;
    *out = &hs;
    return HULK_OK;
}
// NewMinimizerSketch's checks run per read in the reference (minimizer.go:70-76)
int check_host_reads(hulk_ctx *c, const uint64_t *offsets, uint64_t n, uint64_t *max_len_out) {
    uint64_t max_len = 0;
    const uint64_t need = (uint64_t)c->p.w + c->p.k - 1;
    for (uint64_t i = 0; i < n; i++) {
        if (offsets[i + 1] < offsets[i]) return fail(c, HULK_ERR_ARG, "offsets not monotone");
        const uint64_t L = offsets[i + 1] - offsets[i];
        if (L < 1) return fail(c, HULK_ERR_EMPTY_SEQ);
        if (L < need) return fail(c, HULK_ERR_SHORT_SEQ);
        if (L > max_len) max_len = L;
    }
    if (max_len > 0xffffffffull) return fail(c, HULK_ERR_READ_TOO_LONG);
    *max_len_out = max_len;
    return HULK_OK;
}
int ctx_stage_acquire(hulk_ctx *c, size_t nbytes, uint64_t n, StageSet *out) {
    hulk_ctx::HostStage &hs = c->hstage[c->hstage_cur];
    const int rc = stage_ready(c, hs, nbytes, n);
    if (rc != HULK_OK) return rc;
    out->h_bases = hs.h_bases; out->d_bases = hs.d_bases; out->h_off = hs.h_off; out->d_off = hs.d_off; out->cap_bases = hs.cap_bases;
    c->copies_pending = true;                                   // (the caller queues its copies on ctx_stream())
    return HULK_OK;
}
int ctx_device(const hulk_ctx *c) { return c->p.device; }
void ctx_hint_host_offsets(hulk_ctx *c, const uint64_t *h_offsets) { c->h_off_hint = h_offsets; }
int ctx_wait_event(hulk_ctx *c, hipEvent_t e) {
    HIPCHK(c, hipStreamWaitEvent(c->stream, e, 0));
    c->copies_pending = true;                                   // (lane 1 is told through the fork in front of its next batch)
    return HULK_OK;
}
int ctx_record_busy(hulk_ctx *c, hipEvent_t e0, hipEvent_t e1, bool *has1) {
    HIPCHK(c, hipEventRecord(e0, c->stream));
    *has1 = false;
    if (c->lane[1].stream) { HIPCHK(c, hipEventRecord(e1, c->lane[1].stream)); *has1 = true; }
    return HULK_OK;
}
int ctx_stage_release(hulk_ctx *c) {
    hulk_ctx::HostStage &hs = c->hstage[c->hstage_cur];
    { const int rc = stage_mark_busy(c, hs); if (rc != HULK_OK) return rc; }
    c->hstage_cur ^= 1;
    return HULK_OK;
}

}  // namespace hulk
