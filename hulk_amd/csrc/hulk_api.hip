// hulk_api.hip — the C ABI of libhulkhip.so (see include/hulk_hip.h for the reference seam each entry point replaces):
// context creation, the AddSeq variants, Flush / finish, the getters, profiling, `hulk smash`.  Orchestration of a batch is
// hulk_flush.hip, the tables hulk_tables.hip, the multi-GPU entry points hulk_comm.hip (hulk_ctx.h maps the pieces).
#include "hulk_ctx.h"

#include <mutex>

#include <algorithm>
#include <cmath>

using namespace hulk;

namespace {
thread_local std::string g_create_error;

// helpers.Pow (src/helpers/helpers.go:18-28)
uint64_t ipow(uint64_t a, uint64_t b) {
    uint64_t p = 1;
    while (b > 0) { if (b & 1) p *= a; b >>= 1; a *= a; }
    return p;
}

}  // namespace

namespace hulk {
const char *err_text(int status) {
    switch (status) {
        case HULK_OK: return "";
        case HULK_ERR_W: return "w must be: 0 < w < 257";
        case HULK_ERR_K: return "k size must be: 0 < k < 32";
        case HULK_ERR_EMPTY_SEQ: return "sequence length must be > 0";
        case HULK_ERR_SHORT_SEQ: return "sequence length must be >= w + k - 1";
        case HULK_ERR_FEW_BINS: return "not used yet";
        case HULK_ERR_HS_K: return "histosketching only supports k <= 31";
        case HULK_ERR_DECAY: return "decay ratio must be between 0.0 and 1.0";
        case HULK_ERR_BINS: return "histogram must have at least 2 bins";
        case HULK_ERR_NEG_BINS: return "negative value used for number of k-mer spectrum bins";
        case HULK_ERR_NO_SEQ: return "no sequences received";
        case HULK_ERR_FASTQ_ID: return "read ID in fastq file does not begin with @";
        case HULK_ERR_LINE_TOO_LONG: return "bufio.Scanner: token too long";
        case HULK_ERR_IO: return "input/output error";
        case HULK_ERR_FASTA_HEADER: return "fasta input holds no header line";
        case HULK_ERR_ARG: return "invalid argument";
        case HULK_ERR_HIP: return "HIP runtime error";
        case HULK_ERR_NO_DEVICE: return "no usable HIP device (libhulkhip needs an AMD gfx950 GPU)";
        case HULK_ERR_READ_TOO_LONG: return "read longer than the per-read limit of this build";
        case HULK_ERR_STATE: return "call not valid in this state";
        case HULK_ERR_COMM: return "exchange between the ranks failed";
        default: return "unknown error";
    }
}

int fail(hulk_ctx *c, int status, const std::string &extra) {
    std::string msg = err_text(status);
    if (!extra.empty()) msg += ": " + extra;
    if (c) c->last_error = msg; else g_create_error = msg;
    return status;
}
int fail_hip(hulk_ctx *c, hipError_t e, const char *what) {
    return fail(c, HULK_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
}  // namespace hulk

extern "C" {

int hulk_abi_version(void) { return HULK_ABI_VERSION; }
#ifndef HULK_SOURCE_HASH
#define HULK_SOURCE_HASH "unknown"
#endif
#ifndef HULK_HIPCC_VERSION
#define HULK_HIPCC_VERSION "unknown"
#endif
#ifdef HULK_EXPERIMENTS
#define HULK_BUILD_KIND " experiments=1"
#else
#define HULK_BUILD_KIND ""
#endif
const char *hulk_build_info(void) { return "abi=4 arch=gfx950 sources=" HULK_SOURCE_HASH " hipcc=" HULK_HIPCC_VERSION HULK_BUILD_KIND; }
const char *hulk_strerror(int status) { return err_text(status); }
const char *hulk_last_error(const hulk_ctx *ctx) { return ctx ? ctx->last_error.c_str() : g_create_error.c_str(); }

int hulk_create(const hulk_params *params, hulk_ctx **out) {
    if (!out) return fail(nullptr, HULK_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!params) return fail(nullptr, HULK_ERR_ARG, "params is NULL");
    hulk_params p = *params;
    int64_t bins = p.num_bins;
    if (bins == 0) bins = (int64_t)(int32_t)ipow(p.k, 4);           // cmd/sketch.go:118
    // same order as the reference: NewKmerSpectrum (boss.go:57), then NewHistoSketch (sketch.go:277)
    if (bins < 0) return fail(nullptr, HULK_ERR_NEG_BINS, std::to_string(bins));
    if (p.w > 256) return fail(nullptr, HULK_ERR_W);
    if (p.k > 31) return fail(nullptr, HULK_ERR_HS_K);
    if (p.decay_ratio < 0.0 || p.decay_ratio > 1.0 || p.decay_ratio != p.decay_ratio) return fail(nullptr, HULK_ERR_DECAY);
    if (bins < 2) return fail(nullptr, HULK_ERR_BINS);
    if (p.k < 1) return fail(nullptr, HULK_ERR_K);
    // the binning kernels pack (spectrum slot << 20 | bin) into a dword (hulk_spectrum.hip); k^4 <= 31^4 < 2^20
    if (bins > (int64_t)HULK_MAX_BINS)
        return fail(nullptr, HULK_ERR_ARG, "num_bins " + std::to_string(bins) + " exceeds HULK_MAX_BINS (2^20; k^4 at k = 31 is 923521)");
    if (p.flags & ~(HULK_FLAG_GAMMA_CPYTHON | HULK_FLAG_NO_PRUNE | HULK_FLAG_NO_SKIP | HULK_FLAG_SHARD_FULL | HULK_FLAG_NO_OVERLAP | HULK_FLAG_NO_PRERESERVE | HULK_FLAG_CMS_CHAIN))
        return fail(nullptr, HULK_ERR_ARG, "unknown flags");
    if (p.batch > (uint32_t)SCAN_BATCH_MAX) return fail(nullptr, HULK_ERR_ARG, "batch must be 0 (default) or 1..16");
    if (p.work_lanes > 2) return fail(nullptr, HULK_ERR_ARG, "work_lanes must be 0 (default), 1 or 2");
    if (p.reserved != 0) return fail(nullptr, HULK_ERR_ARG, "reserved must be 0");
    if (p.host_copy_threads > 32) return fail(nullptr, HULK_ERR_ARG, "host_copy_threads must be 0 (default) or 1..32");
    if (p.slot_count == 0) { p.slot_begin = 0; p.slot_count = p.sketch_size; }
    if ((uint64_t)p.slot_begin + p.slot_count > p.sketch_size) return fail(nullptr, HULK_ERR_ARG, "slot shard outside sketch");
    if (p.cws_source > HULK_CWS_EXTERNAL) return fail(nullptr, HULK_ERR_ARG, "cws_source");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, HULK_ERR_NO_DEVICE);
    if (p.device < 0 || p.device >= ndev) return fail(nullptr, HULK_ERR_ARG, "device ordinal");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p.device) != hipSuccess) return fail(nullptr, HULK_ERR_NO_DEVICE);
    if (!strstr(prop.gcnArchName, "gfx950"))
        return fail(nullptr, HULK_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName);

    fq_sweep_idle();                                             // (the device FASTQ parser's pooled buffers: hulk_ingest.hip)
    hulk_ctx *c = new hulk_ctx();
    c->p = p; c->B = (int32_t)bins; c->S = p.sketch_size; c->slot_begin = p.slot_begin; c->slots = p.slot_count;
    c->drift = p.decay_ratio != 1.0;
    c->scaling = p.decay_ratio > 0.0 && p.decay_ratio < 1.0;
    c->decay_weight = c->scaling ? std::exp(-p.decay_ratio) : 0.0;   // countmin.go:50-52 (0 otherwise: Go zero value)
    c->cms_width = (int)std::ceil(2 / 0.001);                              // countmin.go:31
    c->cms_depth = (int)std::ceil(std::log(1 - 0.99) / std::log(0.5));     // countmin.go:32
    c->ntiles = (c->B + SCAN_TILE - 1) / SCAN_TILE;
    c->row_stride = (size_t)c->ntiles * SCAN_TILE;

#define CHK_CREATE(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { int rc_ = fail_hip(nullptr, e_, #call); hulk_destroy(c); return rc_; } } while (0)
    CHK_CREATE(hipSetDevice(p.device));
    {   // the count-min replay takes the value a returning LDS atomic hands back as "the counter before this bin": checked on
        // the device, once per process and device; where it does not hold (or on request) the chain-form kernels run
        const int ord = lds_order_verified(p.device);
        if (ord < 0) { const int rc_ = fail_hip(nullptr, (hipError_t)(-ord), "LDS atomic-order self-test"); hulk_destroy(c); return rc_; }
        c->lds_order_ok = ord == 1;
        c->cms_chain = !c->lds_order_ok || (p.flags & HULK_FLAG_CMS_CHAIN) != 0;
    }
    CHK_CREATE(create_lane_stream(&c->own_stream, 0));             // (non-blocking, default priority)
    c->stream = c->own_stream;
    const size_t B = (size_t)c->B, S = c->S, SL = c->slots;
    CHK_CREATE(dalloc(&c->d_state, 1));
    // (the HULK_* variables read here are overrides for profiling scripts; a host sets the fields)
    auto knob = [](const char *name, uint32_t field, uint32_t dflt, uint32_t lo, uint32_t hi) {
        uint32_t v = field ? field : dflt;
        if (const char *e = HULK_EXP_ENV(name)) { const long x = atol(e); if (x >= (long)lo && x <= (long)hi) v = (uint32_t)x; }
        return v;
    };
    c->T = knob("HULK_BATCH", p.batch, SCAN_BATCH_MAX, 1, SCAN_BATCH_MAX);
    // Two work lanes by default — unless count-min decay is on: the replay kernels of that flush hold a CU's whole LDS (112 KB
    // of counters), a second binning lane keeps k_minimizer_fast workgroups (4 x 39.7 KB) resident on every CU twice as much
    // of the time, and the flush, which the next batch on the same ring waits for, no longer finds CUs: C3-shaped
    // 9.3e8 -> 3.0-5.2e8 reads/s with two lanes, C2 +2-4 % (profiles/r04_lanes.txt)
    const bool decay_on = p.decay_ratio > 0.0 && p.decay_ratio < 1.0;
    c->work_lanes = knob("HULK_WORK_LANES", p.work_lanes, decay_on ? 1 : 2, 1, 2);
    c->host_copy_threads = knob("HULK_HOST_COPY_THREADS", p.host_copy_threads, 4, 1, 32);
    c->no_overlap = (p.flags & HULK_FLAG_NO_OVERLAP) != 0 || HULK_EXP_ENV("HULK_NO_OVERLAP") != nullptr;
    c->shard_full = (p.flags & HULK_FLAG_SHARD_FULL) != 0 || HULK_EXP_ENV("HULK_SHARD_FULL") != nullptr;
    c->ring_n = c->T + 1;
    const size_t T = c->T, RN = c->ring_n;
    CHK_CREATE(dalloc(&c->d_hist, 2 * RN * B));
    {   // the flush kernels fill the gaps the VALU-bound minimizer kernels leave: lowest queue priority
        // measured best (7.66e8 reads/s vs 7.59e8 default vs 7.37e8 highest) — the minimizer chain is the
        // critical path
        int lo = 0, hi = 0;
        CHK_CREATE(hipDeviceGetStreamPriorityRange(&lo, &hi));
        const char *pe = HULK_EXP_ENV("HULK_FLUSH_PRIORITY");
        const int prio = pe ? atoi(pe) : lo;   // `lo` = least priority (numerically greatest)
        CHK_CREATE(hipStreamCreateWithPriority(&c->flush_stream, hipStreamNonBlocking, prio));
    }
    CHK_CREATE(hipEventCreateWithFlags(&c->ev_binned, hipEventDisableTiming));
    CHK_CREATE(hipEventCreateWithFlags(&c->ev_flushed[0], hipEventDisableTiming));
    CHK_CREATE(hipEventCreateWithFlags(&c->ev_flushed[1], hipEventDisableTiming));
    CHK_CREATE(dalloc(&c->d_hist_tmp, B));
    CHK_CREATE(dalloc(&c->d_min_slots, (size_t)MIN_SLOTS));
    CHK_CREATE(hipMemsetAsync(c->d_min_slots, 0, (size_t)MIN_SLOTS * 8, c->stream));
    CHK_CREATE(dalloc(&c->d_ctr, (size_t)c->cms_depth * c->cms_width));
    CHK_CREATE(dalloc(&c->d_segsum, T * cms_binorder_entries(c->cms_depth, c->cms_width)));
    CHK_CREATE(dalloc(&c->d_cbase, T * cms_binorder_entries(c->cms_depth, c->cms_width)));
    CHK_CREATE(dalloc(&c->d_f64, T * B));
    CHK_CREATE(dalloc(&c->d_rcp32, T * c->row_stride));
    CHK_CREATE(dalloc(&c->d_rmm, 2 * c->row_stride));
    CHK_CREATE(dalloc(&c->d_mins, S));
    CHK_CREATE(dalloc(&c->d_weights, S));
    CHK_CREATE(dalloc(&c->d_rcb, SL * B * 3));
    CHK_CREATE(dalloc(&c->d_k32, SL * c->row_stride));
    CHK_CREATE(dalloc(&c->d_tilemin, T * ((SL + SCAN_ROWS - 1) / SCAN_ROWS) * SCAN_ROWS * (size_t)c->ntiles * 4));
    CHK_CREATE(dalloc(&c->d_kmin32, (SL ? SL : 1) * (size_t)c->ntiles * 4));
    CHK_CREATE(dalloc(&c->d_rext, T * (size_t)c->ntiles * 4 * 2));
    CHK_CREATE(dalloc(&c->d_kminslot, (SL ? SL : 1)));
    CHK_CREATE(dalloc(&c->d_scanmap, ((SL + SCAN_ROWS - 1) / SCAN_ROWS + 1) * (((size_t)c->ntiles * 4 + 63) / 64)));
    CHK_CREATE(dalloc(&c->d_scanlist, ((SL + SCAN_ROWS - 1) / SCAN_ROWS + 1) * ((size_t)c->ntiles * 4 + 64)));
    CHK_CREATE(dalloc(&c->d_scanlist_n, 4));
    CHK_CREATE(dalloc(&c->d_slotmin, T * ((SL + SCAN_ROWS - 1) / SCAN_ROWS) * SCAN_ROWS));
    CHK_CREATE(dalloc(&c->d_visited, (size_t)MIN_SLOTS));
    CHK_CREATE(hipMemsetAsync(c->d_visited, 0, (size_t)MIN_SLOTS * 8, c->stream));
    // exact pruning of the K scan: without drift weights only fall; with drift (curMin = w / decayWeight) that still
    // holds for negative weights, which is what k_cws_scan tests then; decayRatio == 0 (decayWeight 0) is left alone
    c->prune = !(c->drift && c->decay_weight <= 0.0) && !HULK_EXP_ENV("HULK_NO_PRUNE") && !(p.flags & HULK_FLAG_NO_PRUNE);
    c->no_skip = HULK_EXP_ENV("HULK_NO_SKIP") != nullptr || (p.flags & HULK_FLAG_NO_SKIP) != 0;
    if (c->scaling) {
        const size_t NC = (size_t)c->cms_depth * c->cms_width;
        CHK_CREATE(dalloc(&c->d_blkcnt, T * (size_t)elem_index_blocks(c->B)));
        CHK_CREATE(dalloc(&c->d_eidx, T * B));
        CHK_CREATE(dalloc(&c->d_etot, T));
        CHK_CREATE(dalloc(&c->d_ctrd, NC));
        CHK_CREATE(dalloc(&c->d_segadd, T * cms_binorder_entries(c->cms_depth, c->cms_width)));
        CHK_CREATE(dalloc(&c->d_cstart, T * cms_binorder_entries(c->cms_depth, c->cms_width)));
        CHK_CREATE(dalloc(&c->d_segfac, T * 64));
        CHK_CREATE(dalloc(&c->d_sege0, T * 64));
        CHK_CREATE(hipMemsetAsync(c->d_ctrd, 0, NC * 8, c->stream));
    }
    CHK_CREATE(dalloc(&c->d_candA, T * SL));
    CHK_CREATE(dalloc(&c->d_candB, T * SL));
    CHK_CREATE(hipMemsetAsync(c->d_state, 0, sizeof(DevState), c->stream));
    CHK_CREATE(hipMemsetAsync(c->d_hist, 0, 2 * RN * B * 4, c->stream));
    CHK_CREATE(hipMemsetAsync(c->d_ctr, 0, (size_t)c->cms_depth * c->cms_width * 8, c->stream));
    CHK_CREATE(hipMemsetAsync(c->d_mins, 0, (S ? S : 1) * 8, c->stream));
    CHK_CREATE(launch_fill_f32(c->stream, c->d_rcp32, T * c->row_stride, std::nanf("")));
    {   // weights start at MaxFloat64 (histosketch.go:84-87)
        std::vector<double> w(S ? S : 1, 1.7976931348623157e308);
        CHK_CREATE(hipMemcpy(c->d_weights, w.data(), w.size() * 8, hipMemcpyHostToDevice));
    }
#undef CHK_CREATE
    int rc = build_chains(c);
    if (rc == HULK_OK && p.cws_source == HULK_CWS_GO_COMPAT) rc = generate_tables(c);
    if (rc == HULK_OK) rc = lanes_prereserve(c);
    if (rc == HULK_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = HULK_ERR_HIP;
    if (rc != HULK_OK) { g_create_error = c->last_error.empty() ? err_text(rc) : c->last_error; hulk_destroy(c); return rc; }
    *out = c;
    return HULK_OK;
}

void hulk_destroy(hulk_ctx *c) {
    if (!c) return;
    if (c->stream) hipStreamSynchronize(c->stream);       // the caller's stream may still run our kernels
    if (c->own_stream) hipStreamSynchronize(c->own_stream);
    if (c->flush_stream) { hipStreamSynchronize(c->flush_stream); hipStreamDestroy(c->flush_stream); }
    if (c->ev_binned) hipEventDestroy(c->ev_binned);
    for (int i = 0; i < 2; i++) if (c->ev_flushed[i]) hipEventDestroy(c->ev_flushed[i]);
    for (auto &pr : c->prof) { hipEventDestroy(pr.a); hipEventDestroy(pr.b); }
    for (auto &m : c->marks) hipEventDestroy(m.e);
    hipFree(c->d_state); hipFree(c->d_hist); hipFree(c->d_hist_tmp);
    hipFree(c->d_meta8); hipFree(c->d_segsum); hipFree(c->d_cbase);
    hipFree(c->d_segadd); hipFree(c->d_segfac); hipFree(c->d_cstart); hipFree(c->d_sege0);
    hipFree(c->d_ctr); hipFree(c->d_pos16); hipFree(c->d_mins); hipFree(c->d_f64); hipFree(c->d_weights);
    hipFree(c->d_blkcnt); hipFree(c->d_eidx); hipFree(c->d_etot); hipFree(c->d_ctrd);
    hipFree(c->d_candA); hipFree(c->d_candB); hipFree(c->d_rcb); hipFree(c->d_rcp32); hipFree(c->d_rmm); hipFree(c->d_k32); hipFree(c->d_tilemin);
    hipFree(c->d_scanmap); hipFree(c->d_scanlist); hipFree(c->d_scanlist_n); hipFree(c->d_slotmin); hipFree(c->d_kmin32); hipFree(c->d_rext); hipFree(c->d_visited); hipFree(c->d_kminslot);
    for (auto &hs : c->hstage) {
        if (hs.ev) hipEventDestroy(hs.ev);
        if (hs.ev1) hipEventDestroy(hs.ev1);
        if (hs.h_bases) hipHostFree(hs.h_bases);
        if (hs.h_off) hipHostFree(hs.h_off);
        hipFree(hs.d_bases); hipFree(hs.d_off);
    }
    hipFree(c->d_min_slots);
    for (auto &ln : c->lane) {
        if (ln.stream) { hipStreamSynchronize(ln.stream); hipStreamDestroy(ln.stream); }
        hipFree(ln.d_slow_list); hipFree(ln.d_slow_count);
        hulk::MinimizerList &ml = ln.ml;
        hipFree(ml.x); hipFree(ml.slot); hipFree(ml.key); hipFree(ml.cnt); hipFree(ml.off); hipFree(ml.bsum); hipFree(ml.partial);
        hipFree(ml.nib); hipFree(ml.nib_over); hipFree(ml.lo); hipFree(ml.lo_cnt); hipFree(ml.dmask); hipFree(ml.dsum);
    }
    for (int i = 0; i < 2; i++) if (c->ev_heavy[i]) hipEventDestroy(c->ev_heavy[i]);
    if (c->ev_hold) hipEventDestroy(c->ev_hold);
    if (c->ev_stagger) hipEventDestroy(c->ev_stagger);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    hipFree(c->d_long_xs); hipFree(c->d_long_valid); hipFree(c->d_long_table); hipFree(c->d_long_desc);
    for (auto &H : c->h_long_desc) { if (H.p) hipHostFree(H.p); if (H.ev) hipEventDestroy(H.ev); }
    if (c->ev_long) hipEventDestroy(c->ev_long);
    comm_teardown(c);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
    fq_sweep_idle();
}

int hulk_set_stream(hulk_ctx *c, void *hip_stream) {
    if (!c) return HULK_ERR_ARG;
    { int rcs = sync_all(c); if (rcs != HULK_OK) return rcs; }
    c->stream = (hipStream_t)hip_stream;          // NULL = the HIP null stream
    return HULK_OK;
}

int hulk_set_private_stream(hulk_ctx *c) {
    if (!c) return HULK_ERR_ARG;
    { int rcs = sync_all(c); if (rcs != HULK_OK) return rcs; }
    c->stream = c->own_stream;
    return HULK_OK;
}

int hulk_set_cws_tables(hulk_ctx *c, const double *r, const double *cc, const double *b) {
    if (!c || !r || !cc || !b) return fail(c, HULK_ERR_ARG, "NULL table");
    if (c->seq_count || c->flush_index) return fail(c, HULK_ERR_STATE, "tables must be set before the first read");
    return install_tables(c, r, cc, b);
}

}  // extern "C"

extern "C" {
int hulk_add_reads_device(hulk_ctx *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n,
                          uint32_t max_read_len, uint64_t bases_bytes) {
    if (!c) return HULK_ERR_ARG;
    if (c->finished) return fail(c, HULK_ERR_STATE, "context already finished");
    if (c->sticky != HULK_OK) return fail(c, c->sticky);
    if (n && (!d_bases || !d_offsets)) return fail(c, HULK_ERR_ARG, "NULL buffer");
    const uint64_t I = c->p.interval;
    uint64_t pos = 0;
    const uint64_t *h_off = c->h_off_hint;                         // (ctx_hint_host_offsets: for this call only)
    c->h_off_hint = nullptr;
    struct Clear { hulk_ctx *c; ~Clear() { c->h_off_chunk = nullptr; } } clear_hint{c};
    while (pos < n) {
        c->h_off_chunk = h_off ? h_off + pos : nullptr;
        // one K1 launch covers up to T complete intervals; they are then flushed with ONE pass over K
        const uint64_t fill = I ? c->seq_count % I : 0;
        uint64_t chunk = n - pos;
        if (I) { const uint64_t room = (uint64_t)c->T * I - fill; if (chunk > room) chunk = room; }
        if (chunk > MAX_READS_PER_LAUNCH) chunk = MAX_READS_PER_LAUNCH;   // bounds the minimizer list (HBM)
        int rc = bin_reads(c, d_bases, d_offsets + pos, chunk, max_read_len, bases_bytes, I, fill);
        if (rc != HULK_OK) return rc;
        c->seq_count += chunk; pos += chunk;
        if (I) {                                            // pipeline/sketch.go:211-215
            const uint32_t done = (uint32_t)((fill + chunk) / I);
            rc = flush_batch(c, done);
            if (rc != HULK_OK) return rc;
            c->ring_base = (c->ring_base + done) % c->ring_n;
            if ((fill + chunk) % I == 0) { c->ring_base = 0; if (done) c->cur_ring ^= 1; }  // clean: next batch uses the other ring
        }
    }
    return HULK_OK;
}

int hulk_bin_reads_device_at(hulk_ctx *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n,
                             uint32_t max_read_len, uint64_t bases_bytes, uint64_t reads_per_spectrum, uint32_t first_spectrum) {
    if (!c) return HULK_ERR_ARG;
    { const int rcf = fatal_status(c); if (rcf != HULK_OK) return rcf; }
    if (c->finished) return fail(c, HULK_ERR_STATE, "context already finished");
    if (n && (!d_bases || !d_offsets)) return fail(c, HULK_ERR_ARG, "NULL buffer");
    if (c->ring_base != 0) return fail(c, HULK_ERR_STATE, "a partial interval is pending");
    if (first_spectrum && !reads_per_spectrum) return fail(c, HULK_ERR_ARG, "first_spectrum needs reads_per_spectrum");
    const uint64_t count = reads_per_spectrum ? (n + reads_per_spectrum - 1) / reads_per_spectrum : (n ? 1u : 0u);
    if ((uint64_t)first_spectrum + count > c->T) return fail(c, HULK_ERR_ARG, "more spectra than the batch size");
    const uint64_t skip = (uint64_t)first_spectrum * reads_per_spectrum;       // as if that many reads had been binned before
    for (uint64_t pos = 0; pos < n; pos += MAX_READS_PER_LAUNCH) {
        const uint64_t chunk = std::min<uint64_t>(MAX_READS_PER_LAUNCH, n - pos);
        int rc = bin_reads(c, d_bases, d_offsets + pos, chunk, max_read_len, bases_bytes, reads_per_spectrum, skip + pos, true);
        if (rc != HULK_OK) return rc;
    }
    c->seq_count += n;
    if (n && first_spectrum + (uint32_t)count > c->bin_spectra) c->bin_spectra = first_spectrum + (uint32_t)count;
    return HULK_OK;
}

int hulk_bin_reads_device(hulk_ctx *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n,
                          uint32_t max_read_len, uint64_t bases_bytes, uint64_t reads_per_spectrum) {
    return hulk_bin_reads_device_at(c, d_bases, d_offsets, n, max_read_len, bases_bytes, reads_per_spectrum, 0);
}

int hulk_flush_batch(hulk_ctx *c, uint32_t count) {
    if (!c) return HULK_ERR_ARG;
    if (c->finished) return fail(c, HULK_ERR_STATE, "context already finished");
    if (count > c->T || c->ring_base != 0) return fail(c, HULK_ERR_ARG, "batch count");
    if (count < c->bin_spectra) return fail(c, HULK_ERR_ARG, "fewer spectra flushed than hulk_bin_reads_device filled");
    int rc = flush_batch(c, count);
    if (rc == HULK_OK && count) { c->cur_ring ^= 1; c->bin_spectra = 0; }   // the next batch is binned into the other ring meanwhile
    return rc;
}

int hulk_flush_batch_after(hulk_ctx *c, uint32_t count, void *dep_stream) {
    if (!c) return HULK_ERR_ARG;
    if (c->finished) return fail(c, HULK_ERR_STATE, "context already finished");
    if (count > c->T || c->ring_base != 0) return fail(c, HULK_ERR_ARG, "batch count");
    if (count < c->bin_spectra) return fail(c, HULK_ERR_ARG, "fewer spectra flushed than hulk_bin_reads_device filled");
    int rc = flush_batch(c, count, (hipStream_t)dep_stream, true);
    if (rc == HULK_OK && count) { c->cur_ring ^= 1; c->bin_spectra = 0; }
    return rc;
}

uint32_t hulk_batch_size(const hulk_ctx *c) { return c ? c->T : 0; }

uint32_t *hulk_histogram_device(hulk_ctx *c) { return c ? ring_hist(c) + (size_t)c->ring_base * (size_t)c->B : nullptr; }

int hulk_add_reads(hulk_ctx *c, const uint8_t *bases, const uint64_t *offsets, uint64_t n) {
    if (!c) return HULK_ERR_ARG;
    if (c->finished) return fail(c, HULK_ERR_STATE, "context already finished");
    if (n == 0) return HULK_OK;
    if (!bases || !offsets) return fail(c, HULK_ERR_ARG, "NULL buffer");
    uint64_t max_len = 0;
    { const int rcv = check_host_reads(c, offsets, n, &max_len); if (rcv != HULK_OK) return rcv; }
    // Chunks of <= 2^19 reads / 96 MB go through two pinned staging sets: host copy (several threads: one core
    // copies ~10 GB/s, a PCIe 5 x16 link moves ~50) -> hipMemcpyAsync -> kernels, all queued on the context's
    // stream; the call returns when the caller's buffers have been read, not when the kernels have run.
    constexpr uint64_t CHUNK_READS = 1ull << 19, CHUNK_BYTES = 96ull << 20;
    uint64_t i0 = 0;
    while (i0 < n) {
        uint64_t i1 = i0, cmax = 0;
        while (i1 < n && i1 - i0 < CHUNK_READS && (i1 == i0 || offsets[i1 + 1] - offsets[i0] <= CHUNK_BYTES)) {
            const uint64_t L = offsets[i1 + 1] - offsets[i1];
            if (L > cmax) cmax = L;
            i1++;
        }
        const uint64_t cn = i1 - i0;
        hulk_ctx::HostStage *hsp = nullptr;
        { const int rcs = stage_host_reads(c, bases, offsets, i0, i1, &hsp); if (rcs != HULK_OK) return rcs; }
        hulk_ctx::HostStage &hs = *hsp;
        ctx_hint_host_offsets(c, hs.h_off);                         // (the long-sequence path reads the lengths here)
        const int rc = hulk_add_reads_device(c, hs.d_bases, hs.d_off, cn, (uint32_t)cmax, hs.cap_bases);
        if (rc != HULK_OK) return rc;
        { const int rcb = stage_mark_busy(c, hs); if (rcb != HULK_OK) return rcb; }
        i0 = i1;
    }
    return HULK_OK;
}

int hulk_add_histogram(hulk_ctx *c, const uint32_t *bins) {
    if (!c || !bins) return fail(c, HULK_ERR_ARG, "NULL");
    hipStream_t s = ring_stream(c);                              // (every write to a ring's spectra goes through its lane)
    HIPCHK(c, hipMemcpyAsync(c->d_hist_tmp, bins, (size_t)c->B * 4, hipMemcpyHostToDevice, s));
    { int rcw = ring_ready_for_writes(c); if (rcw != HULK_OK) return rcw; }
    HIPCHK(c, launch_add_hist(s, ring_hist(c) + (size_t)c->ring_base * (size_t)c->B, c->d_hist_tmp, c->B));
    HIPCHK(c, hipStreamSynchronize(s));
    c->hist_hook_used = true;
    return HULK_OK;
}

int hulk_flush(hulk_ctx *c) {
    if (!c) return HULK_ERR_ARG;
    if (c->finished) return fail(c, HULK_ERR_STATE, "context already finished");
    // after hulk_bin_reads_device filled several spectra a flush of ONE would leave the others behind for hulk_finish to
    // flush a second time: they go through hulk_flush_batch[_after]
    if (c->bin_spectra > 1) return fail(c, HULK_ERR_STATE, "hulk_bin_reads_device filled several spectra: use hulk_flush_batch");
    const int rc = flush_batch(c, 1);
    if (rc == HULK_OK) c->bin_spectra = 0;
    return rc;
}

int hulk_finish(hulk_ctx *c) {
    if (!c) return HULK_ERR_ARG;
    { const int rcf = fatal_status(c); if (rcf != HULK_OK) return rcf; }
    if (!c->finished) {
        // pipeline/sketch.go:219-221; a tail batch of hulk_bin_reads_device may span several spectra (ragged last
        // batch of a multi-GPU run): all of them are flushed, in order
        int rc = flush_batch(c, c->bin_spectra > 1 ? c->bin_spectra : 1u);
        if (rc != HULK_OK) return rc;
        c->bin_spectra = 0;
        c->finished = true;
    }
    int rc = check_device_error(c);
    if (rc != HULK_OK) return rc;
    // "no sequences received" (pipeline/sketch.go:237-239); the histogram test hook is exempt — and so is a rank of a sharded
    // run that happened to hold none of a short stream's intervals: the reference's count is over the global stream
    if (c->seq_count == 0 && !c->hist_hook_used && c->comm.global_intervals == 0) return fail(c, HULK_ERR_NO_SEQ);
    return HULK_OK;
}

int hulk_get_sketch(hulk_ctx *c, uint64_t *mins, double *weights) {
    if (!c || !mins || !weights) return fail(c, HULK_ERR_ARG, "NULL");
    { int rcs = sync_all(c); if (rcs != HULK_OK) return rcs; }
    HIPCHK(c, hipMemcpyAsync(mins, c->d_mins, (size_t)c->S * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(weights, c->d_weights, (size_t)c->S * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return HULK_OK;
}

int hulk_get_counters(hulk_ctx *c, uint64_t *n_reads, uint64_t *n_minimizers, uint64_t *total_len) {
    if (!c) return HULK_ERR_ARG;
    DevState st{};
    std::vector<unsigned long long> slots(MIN_SLOTS);
    { int rcs = sync_all(c); if (rcs != HULK_OK) return rcs; }
    HIPCHK(c, hipMemcpyAsync(&st, c->d_state, sizeof st, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(slots.data(), c->d_min_slots, (size_t)MIN_SLOTS * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    unsigned long long nm = 0;
    for (auto v : slots) nm += v;                       // boss.minimizerCounter (boss.go:93)
    if (n_reads) *n_reads = c->seq_count;
    if (n_minimizers) *n_minimizers = nm;
    if (total_len) *total_len = st.total_len;
    return HULK_OK;
}

int hulk_get_histogram(hulk_ctx *c, uint32_t *bins) {
    if (!c || !bins) return fail(c, HULK_ERR_ARG, "NULL");
    { int rcs = sync_all(c); if (rcs != HULK_OK) return rcs; }
    HIPCHK(c, hipMemcpyAsync(bins, ring_hist(c) + (size_t)c->ring_base * (size_t)c->B, (size_t)c->B * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return HULK_OK;
}

int hulk_get_cms(hulk_ctx *c, double *counters) {
    if (!c || !counters) return fail(c, HULK_ERR_ARG, "NULL");
    const size_t n = (size_t)c->cms_depth * c->cms_width;
    { int rcs = sync_all(c); if (rcs != HULK_OK) return rcs; }
    if (c->scaling) {
        HIPCHK(c, hipMemcpyAsync(counters, c->d_ctrd, n * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return HULK_OK;
    }
    std::vector<unsigned long long> tmp(n);
    HIPCHK(c, hipMemcpyAsync(tmp.data(), c->d_ctr, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < n; i++) counters[i] = (double)tmp[i];
    return HULK_OK;
}

int hulk_get_cws_tables(hulk_ctx *c, double *r, double *cc, double *b) {
    if (!c || !r || !cc || !b) return fail(c, HULK_ERR_ARG, "NULL");
    int rc = ensure_tables(c);
    if (rc != HULK_OK) return rc;
    const size_t B = (size_t)c->B;
    std::vector<double> row(B * 3);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (uint32_t s = 0; s < c->slots; s++) {
        HIPCHK(c, hipMemcpy(row.data(), c->d_rcb + (size_t)s * B * 3, B * 3 * 8, hipMemcpyDeviceToHost));
        for (size_t j = 0; j < B; j++) { r[s * B + j] = row[j * 3]; cc[s * B + j] = row[j * 3 + 1]; b[s * B + j] = row[j * 3 + 2]; }
    }
    return HULK_OK;
}

int hulk_selftest_reciprocal(hulk_ctx *c, uint64_t *mismatches) {
    if (!c || !mismatches) return fail(c, HULK_ERR_ARG, "NULL");
    unsigned long long *d = nullptr;
    HIPCHK(c, hipMalloc((void **)&d, 8));
    HIPCHK(c, hipMemsetAsync(d, 0, 8, c->stream));
    HIPCHK(c, launch_selftest_rcp(c->stream, d));
    unsigned long long h = 0;
    HIPCHK(c, hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipFree(d);
    *mismatches = h;
    return HULK_OK;
}

// Device buffers of hulk_smash, kept between calls (grow-only, one set per process and device: three hipMallocs and
// hipFrees were 0.7 of the 2.4 ms a C5 call took end to end).  hulk_smash has no context to hang them on: a mutex
// serialises the calls that share them.
namespace {
struct SmashBuffers {
    std::mutex mu;
    int device = -1;
    unsigned long long *d_m = nullptr; double *d_w = nullptr, *d_o = nullptr, *d_mT = nullptr, *d_wT = nullptr;
    size_t cap_ns = 0, cap_nn = 0, cap_t = 0;
    hipEvent_t ea = nullptr, eb = nullptr;
    void drop() {
        hipFree(d_m); hipFree(d_w); hipFree(d_o); hipFree(d_mT); hipFree(d_wT); d_m = nullptr; d_w = d_o = d_mT = d_wT = nullptr; cap_ns = cap_nn = cap_t = 0;
        if (ea) hipEventDestroy(ea); if (eb) hipEventDestroy(eb); ea = eb = nullptr; device = -1;
    }
};
SmashBuffers g_smash;
}  // namespace

int hulk_smash_ex(int device, const uint64_t *mins, const double *weights, uint32_t n_sketches, uint32_t sketch_size,
                  int metric, double *distances, double *kernel_ms) {
    if (!mins || !weights || !distances) return fail(nullptr, HULK_ERR_ARG, "NULL");
    if (metric != HULK_METRIC_JACCARD && metric != HULK_METRIC_WEIGHTED_JACCARD) return fail(nullptr, HULK_ERR_ARG, "metric");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, HULK_ERR_NO_DEVICE);
    if (device < 0 || device >= ndev) return fail(nullptr, HULK_ERR_ARG, "device ordinal");
    std::lock_guard<std::mutex> lock(g_smash.mu);
    SmashBuffers &B = g_smash;
#define SM_CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { B.drop(); return fail_hip(nullptr, e_, #call); } } while (0)
    const size_t NS = (size_t)n_sketches * sketch_size, NN = (size_t)n_sketches * n_sketches;
    SM_CHK(hipSetDevice(device));
    if (B.device != device) { B.drop(); B.device = device; }
    if (NS > B.cap_ns || !B.d_m) {
        hipFree(B.d_m); hipFree(B.d_w); B.d_m = nullptr; B.d_w = nullptr; B.cap_ns = 0;
        SM_CHK(hipMalloc((void **)&B.d_m, (NS ? NS : 1) * 8));
        SM_CHK(hipMalloc((void **)&B.d_w, (NS ? NS : 1) * 8));
        B.cap_ns = NS;
    }
    const size_t NT = (size_t)smash_padded_n(n_sketches) * sketch_size;
    if (NT > B.cap_t || !B.d_mT) {
        hipFree(B.d_mT); hipFree(B.d_wT); B.d_mT = nullptr; B.d_wT = nullptr; B.cap_t = 0;
        SM_CHK(hipMalloc((void **)&B.d_mT, (NT ? NT : 1) * 8));
        SM_CHK(hipMalloc((void **)&B.d_wT, (NT ? NT : 1) * 8));
        B.cap_t = NT;
    }
    if (NN > B.cap_nn || !B.d_o) {
        hipFree(B.d_o); B.d_o = nullptr; B.cap_nn = 0;
        SM_CHK(hipMalloc((void **)&B.d_o, (NN ? NN : 1) * 8));
        B.cap_nn = NN;
    }
    if (!B.ea) { SM_CHK(hipEventCreate(&B.ea)); SM_CHK(hipEventCreate(&B.eb)); }
    SM_CHK(hipMemcpy(B.d_m, mins, NS * 8, hipMemcpyHostToDevice));
    SM_CHK(hipMemcpy(B.d_w, weights, NS * 8, hipMemcpyHostToDevice));
    if (kernel_ms) SM_CHK(hipEventRecord(B.ea, nullptr));
    SM_CHK(launch_smash(nullptr, B.d_m, B.d_w, n_sketches, sketch_size, metric, B.d_o, B.d_mT, B.d_wT));
    if (kernel_ms) SM_CHK(hipEventRecord(B.eb, nullptr));
    SM_CHK(hipMemcpy(distances, B.d_o, NN * 8, hipMemcpyDeviceToHost));
    if (kernel_ms) { float ms = 0; SM_CHK(hipEventElapsedTime(&ms, B.ea, B.eb)); *kernel_ms = ms; }
#undef SM_CHK
    return HULK_OK;
}

int hulk_release_caches(void) {
    { std::lock_guard<std::mutex> lock(g_smash.mu); g_smash.drop(); }
    hulk::fq_release_idle();
    return HULK_OK;
}

int hulk_smash(int device, const uint64_t *mins, const double *weights, uint32_t n_sketches, uint32_t sketch_size,
               int metric, double *distances) {
    return hulk_smash_ex(device, mins, weights, n_sketches, sketch_size, metric, distances, nullptr);
}

int hulk_get_scan_stats(hulk_ctx *c, uint64_t *tiles_visited, uint64_t *tiles_total) {
    if (!c) return HULK_ERR_ARG;
    { int rcs = sync_all(c); if (rcs != HULK_OK) return rcs; }
    std::vector<unsigned long long> v(MIN_SLOTS);
    HIPCHK(c, hipMemcpyAsync(v.data(), c->d_visited, (size_t)MIN_SLOTS * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    uint64_t sum = 0;
    for (auto x : v) sum += x;
    if (tiles_visited) *tiles_visited = c->prune ? sum : c->scan_tiles_total;
    if (tiles_total) *tiles_total = c->scan_tiles_total;
    return HULK_OK;
}

#ifdef HULK_EXPERIMENTS
int hulk_debug_read(hulk_ctx *c, uint32_t what, void *out, uint64_t *bytes_io) {
    if (!c || !out || !bytes_io) return fail(c, HULK_ERR_ARG, "NULL");
    { int rcs = sync_all(c); if (rcs != HULK_OK) return rcs; }
    const size_t groups = (c->slots + SCAN_ROWS - 1) / SCAN_ROWS, wtiles = (size_t)c->ntiles * 4;
    const void *src = nullptr; size_t need = 0;
    if (what == HULK_DEBUG_TILEMIN) { src = c->d_tilemin; need = groups * wtiles * SCAN_ROWS * sizeof(float); }
    else if (what == HULK_DEBUG_SCANMAP) { src = c->d_scanmap; need = groups * ((wtiles + 63) / 64) * 8; }
    else return fail(c, HULK_ERR_ARG, "hulk_debug_read: what");
    const uint64_t cap = *bytes_io;
    *bytes_io = need;
    if (cap < need) return fail(c, HULK_ERR_ARG, "hulk_debug_read: buffer too small");
    HIPCHK(c, hipMemcpyAsync(out, src, need, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return HULK_OK;
}
#endif

int hulk_get_device_checks(hulk_ctx *c, uint32_t *lds_order_ok, uint32_t *cms_chain_form) {
    if (!c) return HULK_ERR_ARG;
    if (lds_order_ok) *lds_order_ok = c->lds_order_ok ? 1u : 0u;
    if (cms_chain_form) *cms_chain_form = c->cms_chain ? 1u : 0u;
    return HULK_OK;
}

int hulk_synchronize(hulk_ctx *c) {
    if (!c) return HULK_ERR_ARG;
    return sync_all(c);
}

int hulk_set_profiling(hulk_ctx *c, int enabled) {
    if (!c) return HULK_ERR_ARG;
    // 1 (the on/off switch) = every instrumented kernel; otherwise a mask: 2 k_minimizer_fast, 4 k_jump_bin, 8 k_cws_scan, 16 k_cmsd_freq
    c->profiling = enabled == 1 ? 15 : ((enabled & 6) | ((enabled & 8) ? 1 : 0) | ((enabled & 16) ? 8 : 0) | ((enabled & 32) ? 16 : 0));
    return HULK_OK;
}

int hulk_get_profile_table(hulk_ctx *c, char *out, uint64_t cap) {
    if (!c || !out || cap == 0) return fail(c, HULK_ERR_ARG, "NULL");
    { int rcs = sync_all(c); if (rcs != HULK_OK) return rcs; }
    struct Row { const char *k; uint64_t n; double ms; };
    std::vector<Row> rows;
    const size_t M = c->marks.size();
    for (size_t i = 0; i < M; i++) {
        const ProfMark &a = c->marks[i];
        if (a.kernel[0] == '-') continue;
        size_t j = i + 1;
        while (j < M && c->marks[j].s != a.s) j++;                 // the next mark on the same stream ends this kernel
        if (j == M) continue;
        float ms = 0;
        if (hipEventElapsedTime(&ms, a.e, c->marks[j].e) != hipSuccess) continue;
        size_t r = 0;
        while (r < rows.size() && strcmp(rows[r].k, a.kernel) != 0) r++;
        if (r == rows.size()) rows.push_back(Row{a.kernel, 0, 0.0});
        rows[r].n++; rows[r].ms += ms;
    }
    for (auto &m : c->marks) hipEventDestroy(m.e);
    c->marks.clear();
    std::string txt;
    for (const Row &r : rows) { char line[160]; snprintf(line, sizeof line, "%s\t%llu\t%.6f\n", r.k, (unsigned long long)r.n, r.ms); txt += line; }
    if (txt.size() + 1 > cap) return fail(c, HULK_ERR_ARG, "hulk_get_profile_table: buffer too small");
    memcpy(out, txt.c_str(), txt.size() + 1);
    return HULK_OK;
}

int hulk_get_profile(hulk_ctx *c, const char *kernel, uint64_t *launches, double *total_ms) {
    if (!c || !launches || !total_ms) return fail(c, HULK_ERR_ARG, "NULL");
    int which = 0;
    if (kernel && strcmp(kernel, "k_minimizer_fast") == 0) which = 1;
    else if (kernel && strcmp(kernel, "k_jump_bin") == 0) which = 2;
    else if (kernel && strcmp(kernel, "k_jump_left") == 0) which = 3;
    else if (kernel && strcmp(kernel, "k_cmsd_freq") == 0) which = 4;
    else if (kernel && strcmp(kernel, "k_cws_scan") != 0)
        return fail(c, HULK_ERR_ARG, "instrumented kernels: k_cws_scan, k_minimizer_fast, k_jump_bin, k_jump_left, k_cmsd_freq");
    { int rcs = sync_all(c); if (rcs != HULK_OK) return rcs; }
    double tot = 0; uint64_t n = 0;
    std::vector<ProfileRec> keep;
    for (auto &pr : c->prof) {
        if (pr.which != which) { keep.push_back(pr); continue; }
        float ms = 0;
        if (hipEventElapsedTime(&ms, pr.a, pr.b) == hipSuccess) { tot += ms; n++; }
        hipEventDestroy(pr.a); hipEventDestroy(pr.b);
    }
    c->prof.swap(keep);
    *launches = n; *total_ms = tot;
    return HULK_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Multi-GPU: the exchange inside the library (include/hulk_hip.h "multi-GPU with the exchange INSIDE the library").
}  // extern "C"
