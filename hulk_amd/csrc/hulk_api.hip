// hulk_api.hip — the C ABI of libhulkhip.so (see include/hulk_hip.h for the reference seam each
// entry point replaces).  Host-side orchestration only: every numeric step of the path runs in
// the kernels of hulk_minimizer / hulk_spectrum / hulk_countmin / hulk_cws .hip; there is no CPU fallback.
#include "../../include/hulk_hip.h"
#include "hulk_internal.h"
#include "cws_gen.h"
#include "go_rng_jump.h"

#include <rccl/rccl.h>      // types and prototypes only: librccl.so.1 is bound at run time (hulk_comm_init), see Rccl below
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace hulk;

// Debug aid: HULK_POISON=<byte> fills every device / pinned allocation of this file with that byte, so that a read of
// memory nothing has written shows up the same way in every process (tools/fuzz_parity.py found one such read by its
// dependence on what earlier contexts had left behind).
static int poison_byte() {
    static const int v = [] { const char *e = getenv("HULK_POISON"); return e ? (int)(strtol(e, nullptr, 0) & 0xff) : -1; }();
    return v;
}
static hipError_t poison_malloc(void **p, size_t n) {
    hipError_t e = (hipMalloc)(p, n);
    if (e == hipSuccess && poison_byte() >= 0 && n) { e = hipMemset(*p, poison_byte(), n); if (e == hipSuccess) e = hipDeviceSynchronize(); }
    return e;
}
static hipError_t poison_host_malloc(void **p, size_t n, unsigned flags) {
    hipError_t e = (hipHostMalloc)(p, n, flags);
    if (e == hipSuccess && poison_byte() >= 0 && n) memset(*p, poison_byte(), n);
    return e;
}
#define hipMalloc(p, n) poison_malloc((void **)(p), (n))
#define hipHostMalloc(p, n, f) poison_host_malloc((void **)(p), (n), (f))

namespace {

thread_local std::string g_create_error;

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

// helpers.Pow (src/helpers/helpers.go:18-28)
uint64_t ipow(uint64_t a, uint64_t b) {
    uint64_t p = 1;
    while (b > 0) { if (b & 1) p *= a; b >>= 1; a *= a; }
    return p;
}

// go-jump on the host, only for building the static count-min chain tables
int32_t jump_host(uint64_t key, int64_t n) {
    int64_t b = -1, j = 0;
    if (n <= 0) n = 1;
    while (j < n) {
        b = j;
        key = key * 2862933555777941757ull + 1;
        j = (int64_t)((double)(b + 1) * ((double)(1LL << 31) / (double)((key >> 33) + 1)));
    }
    return (int32_t)b;
}

constexpr uint64_t MAX_READS_PER_LAUNCH = 4u << 20;   // 4 Mi reads -> <= ~7 GB of minimizer list at w = 9

struct ProfileRec { hipEvent_t a, b; int which; };   // which: 0 = k_cws_scan, 1 = k_minimizer_fast, 2 = k_jump_bin, 3 = k_jump_left

}  // namespace

struct hulk_ctx {
    hulk_params p{};
    int32_t B = 0;
    uint32_t S = 0, slot_begin = 0, slots = 0;
    int cms_depth = 0, cms_width = 0;
    int ntiles = 0; size_t row_stride = 0;
    bool drift = false, scaling = false;   // ApplyConceptDrift (histosketch.go:79-81), applyScaling (countmin.go:50-55)
    double decay_weight = 0.0;
    uint32_t *d_blkcnt = nullptr, *d_eidx = nullptr, *d_etot = nullptr; double *d_ctrd = nullptr;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // Flushes run on their own stream so that the (memory/latency-bound) count-min + CWS kernels of
    // batch n overlap the (VALU-bound) minimizer kernels of batch n+1.  Two spectrum rings alternate.
    hipStream_t flush_stream = nullptr;
    hipEvent_t ev_binned = nullptr, ev_flushed[2] = {nullptr, nullptr};
    bool pending_flush[2] = {false, false};
    int cur_ring = 0;
    struct PreparedFlush { bool armed = false; FlushBatch fb{}; int ring = 0;
                           bool use_dep = false;        // ev_binned was recorded on a caller's stream (hulk_flush_batch_after)
                           bool allreduce = false;      // hulk_step_sliced: the spectra are summed over the ranks first
    } deferred;   // a flush between its preparation (ev_binned recorded) and the queueing of its kernels
    // the exchange of a multi-rank run (hulk_comm_init*, hulk_step_sharded / hulk_step_sliced)
    struct Comm {
        int kind = 0;                                   // 0 none, 1 RCCL, 2 host callback, 3 loopback
        uint32_t rank = 0, world = 1;
        ncclComm_t nccl = nullptr;
        hulk_exchange_fn fn = nullptr; void *user = nullptr;
        // the collectives run on a stream of their own at the HIGHEST priority: a few workgroups that must not queue behind the
        // thousands of pending minimizer workgroups of the next step (the flush stream around them has the lowest)
        hipStream_t stream = nullptr; hipEvent_t ev_ready = nullptr, ev_done = nullptr;
        uint32_t *d_hdr = nullptr;                      // [world][SHARD_HDR]: {-, need_full, used bins per interval ...}
        uint32_t *d_delta = nullptr;                    // [world][T][depth * width] count-min increments per interval
        uint32_t *d_gather = nullptr; size_t gather_words = 0;   // [world][T][num_bins] spectra of a full exchange
        uint32_t *h_hdr[2] = {nullptr, nullptr}; hipEvent_t ev_hdr[2] = {nullptr, nullptr}; bool hdr_pending[2] = {false, false};
        uint8_t *h_stage = nullptr; size_t h_stage_cap = 0;      // host transport: pinned staging
        unsigned long long *d_sk = nullptr;             // hulk_gather_sketch: [world][2 + 2 S]
        uint64_t step = 0, steps_delta = 0, steps_full = 0, bytes_rx = 0;
        uint64_t global_intervals = 0;                  // intervals of the GLOBAL stream the steps so far covered
    } comm;
    // device state
    DevState *d_state = nullptr;
    uint32_t *d_hist = nullptr, *d_hist_tmp = nullptr;
    unsigned long long *d_ctr = nullptr, *d_mins = nullptr, *d_min_slots = nullptr;
    uint16_t *d_pos16 = nullptr;
    uint8_t *d_meta8 = nullptr; uint32_t *d_segsum = nullptr; unsigned long long *d_cbase = nullptr;   // bin-order count-min
    double *d_segadd = nullptr, *d_segfac = nullptr, *d_cstart = nullptr; uint32_t *d_sege0 = nullptr; // ... with decay
    double *d_f64 = nullptr, *d_weights = nullptr, *d_rcb = nullptr;
    float *d_rcp32 = nullptr, *d_k32 = nullptr, *d_tilemin = nullptr;
    unsigned long long *d_scanmap = nullptr;                                       // [slot groups][wave tiles / 64]: k_scan_test's verdicts
    float *d_slotmin = nullptr;                                                    // [T][slot groups][8]: k_slot_tmin (concept drift only)
    float *d_kmin32 = nullptr, *d_rext = nullptr, *d_kminslot = nullptr;          // bound test of k_cws_scan (no concept drift only)
    unsigned long long *d_visited = nullptr; uint64_t scan_tiles_total = 0; bool prune = false, no_skip = false;
    double *d_candA = nullptr; int32_t *d_candB = nullptr;
    // staging for host reads
    // hulk_add_reads (host buffers): two sets of pinned + device staging; the copy of chunk i+1 into pinned memory
    // and over PCIe runs while the kernels of chunk i do
    struct HostStage {
        uint8_t *h_bases = nullptr, *d_bases = nullptr; uint64_t *h_off = nullptr, *d_off = nullptr;
        size_t cap_bases = 0, cap_off = 0; hipEvent_t ev = nullptr; bool busy = false;
    } hstage[2];
    int hstage_cur = 0;
    uint32_t *d_slow_list = nullptr, *d_slow_count = nullptr; uint64_t d_slow_cap = 0;   // reads the fast kernel deferred (built by k_region_offsets)
    MinimizerList ml{}; uint64_t ml_regions = 0;
    uint64_t *d_long_xs = nullptr, *d_long_table = nullptr; uint8_t *d_long_valid = nullptr;   // long-sequence scratch
    void *d_long_desc = nullptr; uint64_t long_desc_cap = 0;
    uint64_t long_cap = 0, long_table_cap = 0;   // minimizer list of the short-read kernel (grow-only)
    // host-side run state
    uint64_t seq_count = 0, flush_index = 0;
    uint32_t T = 16, ring_n = 17, ring_base = 0;   // interval batch size and spectrum ring
    uint32_t bin_spectra = 0;                      // spectra hulk_bin_reads_device filled that no flush has taken yet
    bool tables_ready = false, finished = false, hist_hook_used = false;
    int sticky = HULK_OK;
    std::string last_error;
    int profiling = 0;   /* bit 0 k_cws_scan, bit 1 k_minimizer_fast, bit 2 k_jump_bin (hulk_set_profiling) */
    std::vector<ProfileRec> prof;
};

namespace {

int fail(hulk_ctx *c, int status, const std::string &extra = std::string()) {
    std::string msg = err_text(status);
    if (!extra.empty()) msg += ": " + extra;
    if (c) c->last_error = msg; else g_create_error = msg;
    return status;
}
int fail_hip(hulk_ctx *c, hipError_t e, const char *what) {
    return fail(c, HULK_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHK(c, call)                                             \
    do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail_hip((c), e_, #call); } while (0)

template <typename T> hipError_t dalloc(T **p, size_t n) { return hipMalloc((void **)p, n ? n * sizeof(T) : sizeof(T)); }

// static chain tables for the count-min prefix sums: for row d, bins grouped by counter
// position g = jump(bin*(d+1), width) (countmin.go:122-125), ascending bin inside a group.
int build_chains(hulk_ctx *c) {
    const int D = c->cms_depth, W = c->cms_width; const int32_t B = c->B;
    if (!getenv("HULK_CHAINS_HOST")) {          // (the host loop below is kept as the A/B check of k_build_chains)
        HIPCHK(c, dalloc(&c->d_meta8, (size_t)D * B));
        HIPCHK(c, dalloc(&c->d_pos16, (size_t)D * B));
        HIPCHK(c, launch_build_chains(c->stream, c->d_pos16, c->d_meta8, B, D, W));
        return HULK_OK;
    }
    std::vector<uint32_t> pos(B);
    std::vector<uint16_t> pos16((size_t)D * B);
    std::vector<uint8_t> meta8((size_t)D * B);       // bits 0-6: previous lane of the 64-bin chunk on the same counter (64 = none); bit 7: last one
    for (int d = 0; d < D; d++) {
        for (int32_t b = 0; b < B; b++) {
            uint64_t h = (uint64_t)b + (uint64_t)d * (uint64_t)b;      // countmin.go:123-125: hash(bin + d * bin)
            pos[b] = (uint32_t)jump_host(h, W);
            pos16[(size_t)d * B + b] = (uint16_t)pos[b];
        }
        std::vector<int32_t> last_bin(W, -1);
        for (int32_t c0 = 0; c0 < B; c0 += 64) {
            const int32_t c1 = std::min<int32_t>(B, c0 + 64);
            for (int32_t b = c0; b < c1; b++) {
                const int32_t prev = last_bin[pos[b]];
                uint8_t m = 64;
                if (prev >= c0) { m = (uint8_t)(prev - c0); meta8[(size_t)d * B + prev] &= 0x7f; }   // prev is no longer the last
                meta8[(size_t)d * B + b] = m | 0x80;
                last_bin[pos[b]] = b;
            }
        }
    }
    HIPCHK(c, dalloc(&c->d_meta8, meta8.size()));
    HIPCHK(c, hipMemcpy(c->d_meta8, meta8.data(), meta8.size(), hipMemcpyHostToDevice));
    HIPCHK(c, dalloc(&c->d_pos16, pos16.size()));
    HIPCHK(c, hipMemcpy(c->d_pos16, pos16.data(), pos16.size() * 2, hipMemcpyHostToDevice));
    return HULK_OK;
}

// upload r,c,b rows owned by this context as interleaved {r,c,b} and derive the fp32 K table
int install_tables(hulk_ctx *c, const double *r, const double *cc, const double *b) {
    const size_t B = (size_t)c->B;
    std::vector<double> row(B * 3);
    for (uint32_t s = 0; s < c->slots; s++) {
        const size_t src = (size_t)(c->slot_begin + s) * B;
        for (size_t j = 0; j < B; j++) { row[j * 3] = r[src + j]; row[j * 3 + 1] = cc[src + j]; row[j * 3 + 2] = b[src + j]; }
        HIPCHK(c, hipMemcpy(c->d_rcb + (size_t)s * B * 3, row.data(), B * 3 * sizeof(double), hipMemcpyHostToDevice));
    }
    HIPCHK(c, launch_build_k32(c->stream, c->d_rcb, c->d_k32, (int)c->slots, c->B, c->row_stride));
    if (c->slots) {
        HIPCHK(c, launch_tile_kmin(c->stream, c->d_k32, c->d_kmin32, (int)c->slots, c->ntiles, c->row_stride));
        HIPCHK(c, launch_slot_kmin(c->stream, c->d_kmin32, c->d_kminslot, (int)c->slots, c->ntiles));
    }
    c->tables_ready = true;
    return HULK_OK;
}

// newCWS (histosketch.go:95-126): the host walks Go's math/rand streams (cws_gen.h), the device does
// the gamma math and the in-order compaction, chunk by chunk (double-buffered pinned staging).
int generate_tables_host(hulk_ctx *c) {
    const uint64_t B = (uint64_t)c->B;
    const uint64_t need_entries = (uint64_t)(c->slot_begin + c->slots) * B;     // rows of later slots are not needed
    if (need_entries == 0) { c->tables_ready = true; return HULK_OK; }
    const uint64_t need_gammas = 2 * need_entries;
    const size_t CH = (size_t)1 << 22;                                        // attempts (or uniforms) per chunk
    const CwsConstants K((c->p.flags & HULK_FLAG_GAMMA_CPYTHON) != 0);
    uint64_t *h_buf[2] = {nullptr, nullptr}; uint64_t *d_pairs[2] = {nullptr, nullptr};
    double *d_val = nullptr; uint32_t *d_blkcnt = nullptr; unsigned long long *d_tot = nullptr;
    hipEvent_t done[2] = {nullptr, nullptr};
    int rc = HULK_OK;
    auto cleanup = [&]() {
        for (int i = 0; i < 2; i++) { if (h_buf[i]) hipHostFree(h_buf[i]); hipFree(d_pairs[i]); if (done[i]) hipEventDestroy(done[i]); }
        hipFree(d_val); hipFree(d_blkcnt); hipFree(d_tot);
    };
#define GEN_CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = fail_hip(c, e_, #call); cleanup(); return rc; } } while (0)
    for (int i = 0; i < 2; i++) {
        GEN_CHK(hipHostMalloc((void **)&h_buf[i], CH * 16, hipHostMallocDefault));
        GEN_CHK(hipMalloc((void **)&d_pairs[i], CH * 16));
        GEN_CHK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
    }
    GEN_CHK(hipMalloc((void **)&d_val, CH * 8));
    GEN_CHK(hipMalloc((void **)&d_blkcnt, (CH / 1024 + 1) * 4));
    GEN_CHK(hipMalloc((void **)&d_tot, 16));
    GEN_CHK(hipMemsetAsync(d_tot, 0, 16, c->stream));
    // ---- r and c: gamma variates
    {
        AttemptStream attempts;
        unsigned long long got = 0; int cur = 0; uint64_t inflight[2] = {0, 0};
        // acceptance of Cheng's sampler at alpha = 2 is ~0.8; the tail chunk is sized from the estimate
        while (got < need_gammas) {
            uint64_t want = (uint64_t)((double)(need_gammas - got) / 0.78) + 4096;
            if (want > CH) want = CH;
            if (inflight[cur]) GEN_CHK(hipEventSynchronize(done[cur]));         // staging buffer free again
            attempts.fill(h_buf[cur], (size_t)want);
            GEN_CHK(hipMemcpyAsync(d_pairs[cur], h_buf[cur], want * 16, hipMemcpyHostToDevice, c->stream));
            GEN_CHK(launch_cws_chunk(c->stream, d_pairs[cur], want, d_val, d_blkcnt, d_tot, d_tot + 1, c->d_rcb, B,
                                     c->slot_begin, c->slots, c->S, K.ainv, K.bbb, K.ccc, K.magic, nullptr, 0, nullptr, 0));
            GEN_CHK(hipEventRecord(done[cur], c->stream));
            inflight[cur] = want;
            cur ^= 1;
            // progress is only needed near the end; until then overlap host generation with the device
            if ((double)(got + (unsigned long long)(0.70 * (double)want)) >= (double)need_gammas || want < CH) {
                GEN_CHK(hipMemcpyAsync(&got, d_tot, 8, hipMemcpyDeviceToHost, c->stream));
                GEN_CHK(hipStreamSynchronize(c->stream));
            } else {
                got += (unsigned long long)(0.70 * (double)want);                // safe under-estimate
            }
        }
    }
    // ---- b = U(0,1) * r
    {
        UniformStream uni;
        int cur = 0; bool used[2] = {false, false};
        for (uint64_t first = 0; first < need_entries; first += CH) {
            const uint64_t n = std::min<uint64_t>(CH, need_entries - first);
            if (used[cur]) GEN_CHK(hipEventSynchronize(done[cur]));
            uni.fill(h_buf[cur], (size_t)n);
            GEN_CHK(hipMemcpyAsync(d_pairs[cur], h_buf[cur], n * 8, hipMemcpyHostToDevice, c->stream));
            GEN_CHK(launch_cws_beta(c->stream, d_pairs[cur], first, n, c->d_rcb, B, c->slot_begin, c->slots));
            GEN_CHK(hipEventRecord(done[cur], c->stream));
            used[cur] = true; cur ^= 1;
        }
    }
    GEN_CHK(launch_build_k32(c->stream, c->d_rcb, c->d_k32, (int)c->slots, c->B, c->row_stride));
    if (c->slots) {
        GEN_CHK(launch_tile_kmin(c->stream, c->d_k32, c->d_kmin32, (int)c->slots, c->ntiles, c->row_stride));
        GEN_CHK(launch_slot_kmin(c->stream, c->d_kmin32, c->d_kminslot, (int)c->slots, c->ntiles));
    }
    GEN_CHK(hipStreamSynchronize(c->stream));
#undef GEN_CHK
    cleanup();
    c->tables_ready = true;
    return HULK_OK;
}

// The same tables with the math/rand stream generated ON THE DEVICE (k_alfg_jump / k_alfg_fill: chunks of 2^20
// values started in parallel through the jump polynomial of go_rng_jump.h).  Both go_rng generators are seeded with
// 1, so ONE raw stream serves the gamma attempts (two values each) and the uniforms (one per entry).  The only
// data-dependent part of the consumption — an attempt whose u1 fails the range test takes one value instead of two,
// 2e-7 of them — is found by k_rng_candidates and resolved here into the `ev` list k_cws_eval uses; a value that
// would make Float64() resample (2^-54) sends the whole generation to the host walk instead.
// returns HULK_OK, an error, or +1 = "use the host generator".
int generate_tables_device(hulk_ctx *c) {
    const uint64_t B = (uint64_t)c->B;
    const uint64_t need_entries = (uint64_t)(c->slot_begin + c->slots) * B;
    if (need_entries == 0) { c->tables_ready = true; return HULK_OK; }
    const uint64_t need_gammas = 2 * need_entries;
    const uint64_t C = 1ull << GO_RNG_JUMP_LOG2;
    const size_t CH = (size_t)1 << 22;                                        // attempts per evaluation chunk
    const CwsConstants K((c->p.flags & HULK_FLAG_GAMMA_CPYTHON) != 0);
    // Cheng's sampler accepts ~77 % of the attempts at alpha = 2; the stream is sized with a wide margin
    const uint64_t max_attempts = (uint64_t)((double)need_gammas / 0.66) + (1u << 20);
    const uint64_t n_chunks = (2 * max_attempts + 4096 + C - 1) / C;
    const uint64_t total_raw = n_chunks * C;
    uint64_t *d_raw = nullptr, *d_win = nullptr, *d_coef = nullptr, *d_list = nullptr, *d_ev = nullptr;
    unsigned int *d_cnt = nullptr; double *d_val = nullptr; uint32_t *d_blkcnt = nullptr; unsigned long long *d_tot = nullptr;
    const uint32_t LIST_CAP = 1u << 16;
    int rc = HULK_OK;
    auto cleanup = [&]() { hipFree(d_raw); hipFree(d_win); hipFree(d_coef); hipFree(d_list); hipFree(d_ev); hipFree(d_cnt);
                           hipFree(d_val); hipFree(d_blkcnt); hipFree(d_tot); };
#define GEN_CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = fail_hip(c, e_, #call); cleanup(); return rc; } } while (0)
    if (hipMalloc((void **)&d_raw, total_raw * 8) != hipSuccess) { (void)hipGetLastError(); cleanup(); return 1; }   // not enough HBM: host walk
    GEN_CHK(hipMalloc((void **)&d_win, n_chunks * 607 * 8));
    GEN_CHK(hipMalloc((void **)&d_coef, 2 * 607 * 8));                         // x^(2^20) and x^(2^26)
    GEN_CHK(hipMalloc((void **)&d_list, (size_t)LIST_CAP * 8));
    GEN_CHK(hipMalloc((void **)&d_cnt, 4));
    GEN_CHK(hipMalloc((void **)&d_val, CH * 8));
    GEN_CHK(hipMalloc((void **)&d_blkcnt, (CH / 1024 + 1) * 4));
    GEN_CHK(hipMalloc((void **)&d_tot, 16));
    {
        uint64_t w0[607];
        GoRandSource(1).initial_window(w0);
        GEN_CHK(hipMemcpyAsync(d_win, w0, sizeof w0, hipMemcpyHostToDevice, c->stream));
        GEN_CHK(hipMemcpyAsync(d_coef, GO_RNG_JUMP, 607 * 8, hipMemcpyHostToDevice, c->stream));
        GEN_CHK(hipMemcpyAsync(d_coef + 607, GO_RNG_JUMP_FAR, 607 * 8, hipMemcpyHostToDevice, c->stream));
        GEN_CHK(hipMemsetAsync(d_cnt, 0, 4, c->stream));
        GEN_CHK(hipMemsetAsync(d_tot, 0, 16, c->stream));
        GEN_CHK(hipStreamSynchronize(c->stream));                              // w0 is a stack buffer
    }
    static const bool one_level = getenv("HULK_ALFG_ONE_LEVEL") != nullptr;      // A/B aid: the single walk over all chunks
    GEN_CHK(launch_alfg(c->stream, d_coef, one_level ? nullptr : d_coef + 607, 1u << (GO_RNG_JUMP_FAR_LOG2 - GO_RNG_JUMP_LOG2),
                        d_win, d_raw, 0, (uint32_t)n_chunks, C));
    GEN_CHK(launch_rng_candidates(c->stream, d_raw, total_raw, d_list, LIST_CAP, d_cnt));
    unsigned int n_cand = 0;
    GEN_CHK(hipMemcpyAsync(&n_cand, d_cnt, 4, hipMemcpyDeviceToHost, c->stream));
    GEN_CHK(hipStreamSynchronize(c->stream));
    if (n_cand > LIST_CAP) { cleanup(); return 1; }
    std::vector<uint64_t> cand(n_cand), ev;
    if (n_cand) GEN_CHK(hipMemcpy(cand.data(), d_list, (size_t)n_cand * 8, hipMemcpyDeviceToHost));
    std::sort(cand.begin(), cand.end());
    {   // walk the candidates: p0 = stream position of the u1 of valid attempt i0
        uint64_t p0 = 0, i0 = 0;
        for (uint64_t cd : cand) {
            const uint64_t pos = cd >> 1;
            if (cd & 1) { cleanup(); return 1; }                               // Float64() would resample here: host walk
            if (pos < p0 || ((pos - p0) & 1)) continue;                        // a u2 position: no range test there
            const uint64_t k = i0 + (pos - p0) / 2;                            // the valid attempt that follows the dead one
            ev.push_back(k);
            p0 = pos + 1; i0 = k;
        }
    }
    if (!ev.empty()) {
        GEN_CHK(hipMalloc((void **)&d_ev, ev.size() * 8));
        GEN_CHK(hipMemcpy(d_ev, ev.data(), ev.size() * 8, hipMemcpyHostToDevice));
    }
    // ---- r and c: gamma variates, in chunks of CH attempts straight from the device stream
    {
        unsigned long long got = 0; uint64_t next_attempt = 0;
        while (got < need_gammas) {
            uint64_t plan = (uint64_t)((double)(need_gammas - got) / 0.80) + 4096;   // a slight under-estimate: no overshoot of chunks
            while (plan > 0) {
                const uint64_t want = std::min<uint64_t>(plan, CH);
                if (2 * (next_attempt + want) + ev.size() + 2 > total_raw) { cleanup(); return 1; }   // margin exhausted (not expected)
                GEN_CHK(launch_cws_chunk(c->stream, nullptr, want, d_val, d_blkcnt, d_tot, d_tot + 1, c->d_rcb, B,
                                         c->slot_begin, c->slots, c->S, K.ainv, K.bbb, K.ccc, K.magic, d_raw, next_attempt,
                                         d_ev, (uint32_t)ev.size()));
                next_attempt += want; plan -= want;
            }
            GEN_CHK(hipMemcpyAsync(&got, d_tot, 8, hipMemcpyDeviceToHost, c->stream));
            GEN_CHK(hipStreamSynchronize(c->stream));
        }
    }
    // ---- b = U(0,1) * r: entry e takes stream value e (the uniform generator is a second source with the same seed)
    GEN_CHK(launch_cws_beta(c->stream, d_raw, 0, need_entries, c->d_rcb, B, c->slot_begin, c->slots));
    GEN_CHK(launch_build_k32(c->stream, c->d_rcb, c->d_k32, (int)c->slots, c->B, c->row_stride));
    if (c->slots) {
        GEN_CHK(launch_tile_kmin(c->stream, c->d_k32, c->d_kmin32, (int)c->slots, c->ntiles, c->row_stride));
        GEN_CHK(launch_slot_kmin(c->stream, c->d_kmin32, c->d_kminslot, (int)c->slots, c->ntiles));
    }
    GEN_CHK(hipStreamSynchronize(c->stream));
#undef GEN_CHK
    cleanup();
    c->tables_ready = true;
    return HULK_OK;
}

int generate_tables(hulk_ctx *c) {
    static const bool host_only = getenv("HULK_CWS_HOST") != nullptr;
    if (!host_only) {
        const int rc = generate_tables_device(c);
        if (rc <= 0) return rc;                                                // done, or a real error
    }
    return generate_tables_host(c);
}

int ensure_tables(hulk_ctx *c) {
    if (c->tables_ready) return HULK_OK;
    if (c->p.cws_source == HULK_CWS_EXTERNAL)
        return fail(c, HULK_ERR_STATE, "cws_source is EXTERNAL but hulk_set_cws_tables was not called");
    return generate_tables(c);
}

uint32_t *ring_hist(hulk_ctx *c) { return c->d_hist + (size_t)c->cur_ring * (size_t)c->ring_n * (size_t)c->B; }

// the work stream may only write spectra of the current ring once the flush that last read them is done
int issue_flush(hulk_ctx *c, hipEvent_t gate);   // (defined below)
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
    if (hipEvent_t e = ring_write_event(c)) HIPCHK(c, hipStreamWaitEvent(c->stream, e, 0));
    return HULK_OK;
}

// The profile events only measure time (hulk_get_profile synchronises the streams before it reads them): without the
// system-scope fence a default event carries, a bracket no longer writes back and invalidates the caches around the kernel
// (k_minimizer_fast's bracket cost 2 % of a C2 step that way, mostly in the kernel behind it)
constexpr unsigned PROFILE_EVENT_FLAGS = hipEventDisableSystemFence;

int sync_all(hulk_ctx *c) {
    { const int rc = issue_flush(c, nullptr); if (rc != HULK_OK) return rc; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
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
int bin_long_reads(hulk_ctx *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n, MinimizerParams P,
                   uint32_t *hist) {
    std::vector<uint64_t> off(n + 1);
    HIPCHK(c, hipMemcpyAsync(off.data(), d_offsets, (n + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // groups of long sequences, one launch set per group: bounded scratch (positions) and grid.y
    constexpr uint64_t GROUP_POS = 128ull << 20;        // positions per group (8 B + 1 B scratch, <= 16 B of table each)
    constexpr uint32_t GROUP_SEQS = 32768;
    std::vector<hulk::LongSeqDesc> descs;
    uint64_t pos_total = 0, tab_total = 0, max_npos = 0;
    auto launch_group = [&]() -> int {
        if (descs.empty()) return HULK_OK;
        if (pos_total > c->long_cap) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            hipFree(c->d_long_xs); hipFree(c->d_long_valid); c->d_long_xs = nullptr; c->d_long_valid = nullptr; c->long_cap = 0;
            HIPCHK(c, hipMalloc((void **)&c->d_long_xs, pos_total * 8));
            HIPCHK(c, hipMalloc((void **)&c->d_long_valid, pos_total));
            c->long_cap = pos_total;
        }
        if (tab_total > c->long_table_cap) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            hipFree(c->d_long_table); c->d_long_table = nullptr; c->long_table_cap = 0;
            HIPCHK(c, hipMalloc((void **)&c->d_long_table, tab_total * 8));
            c->long_table_cap = tab_total;
        }
        if (descs.size() > c->long_desc_cap) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            hipFree(c->d_long_desc); c->d_long_desc = nullptr; c->long_desc_cap = 0;
            HIPCHK(c, hipMalloc((void **)&c->d_long_desc, (descs.size() + 1024) * sizeof(hulk::LongSeqDesc)));
            c->long_desc_cap = descs.size() + 1024;
        }
        // pageable source: the copy is staged before the call returns, descs may be reused afterwards
        HIPCHK(c, hipMemcpyAsync(c->d_long_desc, descs.data(), descs.size() * sizeof(hulk::LongSeqDesc),
                                 hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, launch_long_group(c->stream, d_bases, (const hulk::LongSeqDesc *)c->d_long_desc, (uint32_t)descs.size(),
                                    max_npos, P, c->d_long_xs, c->d_long_valid, c->d_long_table, tab_total, hist,
                                    c->d_min_slots));
        HIPCHK(c, hipStreamSynchronize(c->stream));      // descs.data() is pageable memory: keep it simple and ordered
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

int bin_reads(hulk_ctx *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n,
              uint32_t max_len, uint64_t bases_bytes, uint64_t interval, uint64_t fill) {
    MinimizerParams P{};
    P.k = c->p.k; P.w = c->p.w; P.num_bins = c->B; P.bases_bytes = bases_bytes;
    P.interval = interval; P.fill = fill; P.ring_base = c->ring_base; P.ring_n = c->ring_n;
    // whole intervals in front of this launch move the first spectrum, not the fill: the kernels then see a launch that
    // starts inside spectrum ring_base (hist_slot() is unchanged by this) and build no empty spectra in front of it
    if (P.interval && P.fill >= P.interval) { P.ring_base = (uint32_t)((P.ring_base + P.fill / P.interval) % P.ring_n); P.fill %= P.interval; }
    if (const char *e = getenv("HULK_K1_DEBUG")) P.debug = (uint32_t)atoi(e);
    { int rcw = ring_issue_own_flush(c); if (rcw != HULK_OK) return rcw; }
    uint32_t *hist = ring_hist(c);
    int threads = 256;
    // the short-read kernel takes reads of <= 16*w k-mer positions and <= 256 bases; when the batch's
    // length bound already exceeds that, go straight to the generic kernel
    // ... or, two groups per read, <= 2*16w - (w-1) positions (300 bases at k = 21, w = 9) while a group's own
    // 16w + k - 1 bases fit its 256-base staging
    const bool fast_base = c->p.w >= 1 && c->p.w <= 16 && !getenv("HULK_NO_FAST_K1") && n < 0xffffffffull;
    const bool single_ok = max_len <= 256 && (uint64_t)max_len < (uint64_t)c->p.k + 16ull * c->p.w;
    const bool pair_ok = !single_ok && !getenv("HULK_NO_PAIR") && 16ull * c->p.w + c->p.k - 1 <= 256 && max_len <= 512 &&
                         (uint64_t)max_len < (uint64_t)c->p.k + 32ull * c->p.w - (c->p.w - 1);
    const bool fast_ok = fast_base && (single_ok || pair_ok);
    P.pair = pair_ok ? 1u : 0u;
    if (fast_ok) {
        // short-read kernel first; reads it cannot take (N bases, too long for 16 blocks of w
        // positions) are queued on the device and binned by the generic kernel right after
        if (n > c->d_slow_cap) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            hipFree(c->d_slow_list); c->d_slow_list = nullptr; c->d_slow_cap = 0;
            HIPCHK(c, hipMalloc((void **)&c->d_slow_list, (size_t)(n + n / 4 + 1024) * 4));
            c->d_slow_cap = n + n / 4 + 1024;
        }
        const uint64_t regions = (n + FAST_READS_PER_WAVE - 1) / FAST_READS_PER_WAVE;
        // (a region never shrinks again: calls with and without reads of two groups may alternate)
        const uint64_t rcap = std::max<uint64_t>(minimizer_list_rcap(c->p.w, pair_ok), c->ml.rcap);
        if (regions > c->ml_regions || c->ml.rcap != rcap) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            hipFree(c->ml.x); hipFree(c->ml.slot); hipFree(c->ml.key); hipFree(c->ml.cnt); hipFree(c->ml.off); hipFree(c->ml.bsum);
            hipFree(c->ml.lo); hipFree(c->ml.lo_cnt); hipFree(c->ml.dmask); hipFree(c->ml.dsum);
            uint32_t *keep_partial = c->ml.partial; const uint32_t keep_parts = c->ml.max_parts;
            uint32_t *keep_nib = c->ml.nib, *keep_over = c->ml.nib_over; const uint32_t keep_np = c->ml.nib_parts;
            c->ml = MinimizerList{}; c->ml_regions = 0;
            c->ml.partial = keep_partial; c->ml.max_parts = keep_parts;
            c->ml.nib = keep_nib; c->ml.nib_over = keep_over; c->ml.nib_parts = keep_np;
            const uint64_t cap = regions + regions / 8 + 64;
            HIPCHK(c, hipMalloc((void **)&c->ml.x, cap * rcap * 8));
            HIPCHK(c, hipMalloc((void **)&c->ml.slot, cap * rcap));
            HIPCHK(c, hipMalloc((void **)&c->ml.key, cap * rcap * 4));
            HIPCHK(c, hipMalloc((void **)&c->ml.cnt, cap * 4));
            HIPCHK(c, hipMalloc((void **)&c->ml.off, (cap + 1) * 4));
            HIPCHK(c, hipMalloc((void **)&c->ml.bsum, (cap / 1024 + 2) * 4));
            HIPCHK(c, hipMalloc((void **)&c->ml.lo, cap * JUMP_LO_CAP * sizeof(uint4)));
            HIPCHK(c, hipMalloc((void **)&c->ml.lo_cnt, cap * 4));
            HIPCHK(c, hipMalloc((void **)&c->ml.dmask, cap * 4));
            HIPCHK(c, hipMalloc((void **)&c->ml.dsum, (cap / 1024 + 2) * 4));
            if (!c->ml.nib) {
                const size_t nr = ((size_t)c->B + 262143) / 262144;
                c->ml.nib_parts = 48;
                HIPCHK(c, hipMalloc((void **)&c->ml.nib, (size_t)c->ml.nib_parts * c->ring_n * nr * (262144 / 8) * 4));
                HIPCHK(c, hipMalloc((void **)&c->ml.nib_over, RING_MAX * 4));
                HIPCHK(c, hipMemsetAsync(c->ml.nib_over, 0, RING_MAX * 4, c->stream));
            }
            if (!c->ml.partial) {
                c->ml.max_parts = 8;
                HIPCHK(c, hipMalloc((void **)&c->ml.partial, (size_t)c->ml.max_parts * c->ring_n * (size_t)c->B * 4));
            }
            c->ml.rcap = rcap; c->ml_regions = cap;
        }
        ProfileRec pr{}; pr.which = 1;
        if ((c->profiling & 2)) {
            HIPCHK(c, hipEventCreateWithFlags(&pr.a, PROFILE_EVENT_FLAGS)); HIPCHK(c, hipEventCreateWithFlags(&pr.b, PROFILE_EVENT_FLAGS));
            HIPCHK(c, hipEventRecord(pr.a, c->stream));
        }
        HIPCHK(c, launch_minimizer_fast(c->stream, d_bases, d_offsets, n, P, c->ml, c->d_state, c->d_min_slots));
        if ((c->profiling & 2)) { HIPCHK(c, hipEventRecord(pr.b, c->stream)); c->prof.push_back(pr); }
        ProfileRec pj{}; pj.which = 2;
        ProfileRec pl{}; pl.which = 3;
        if ((c->profiling & 4)) {
            HIPCHK(c, hipEventCreateWithFlags(&pj.a, PROFILE_EVENT_FLAGS)); HIPCHK(c, hipEventCreateWithFlags(&pj.b, PROFILE_EVENT_FLAGS));
            HIPCHK(c, hipEventCreateWithFlags(&pl.a, PROFILE_EVENT_FLAGS)); HIPCHK(c, hipEventCreateWithFlags(&pl.b, PROFILE_EVENT_FLAGS));
        }
        // (the minimizer and jump-hash kernels do not touch the spectra: only the histogram kernels behind them wait for
        // the flush that last read this ring)
        HIPCHK(c, launch_minimizer_post(c->stream, n, P, c->ml, hist, c->d_slow_list, c->d_slow_count, pj.a, pj.b,
                                        ring_write_event(c), pl.a, pl.b));
        if ((c->profiling & 4)) { c->prof.push_back(pj); c->prof.push_back(pl); }
        pick_config(c->p.k, max_len, P, threads);      // (fast_ok implies max_len <= 256: always fits)
        // the list is normally empty or short; its length is only known on the device, so the grid is fixed: enough
        // workgroups that 1 % of deferred reads (reads with N) do not queue behind 512 waves (blocks past the list exit at once)
        static const uint32_t slow_blocks = [] { const char *e = getenv("HULK_SLOW_BLOCKS"); const long v = e ? atol(e) : 2048; return (uint32_t)(v < 1 ? 1 : v > 8192 ? 8192 : v); }();
        const uint32_t list_blocks = (uint32_t)std::min<uint64_t>(slow_blocks, (n + 3) / 4);
        HIPCHK(c, launch_minimizer_bin(c->stream, d_bases, d_offsets, n, P, threads, hist, c->d_state,
                                       c->d_min_slots, c->d_slow_list, c->d_slow_count, list_blocks));
        return HULK_OK;
    }
    { const int rcf = issue_flush(c, nullptr); if (rcf != HULK_OK) return rcf; }
    if (hipEvent_t e = ring_write_event(c)) HIPCHK(c, hipStreamWaitEvent(c->stream, e, 0));
    const bool fits = pick_config(c->p.k, max_len, P, threads);
    P.skip_long = fits ? 0u : 1u;
    HIPCHK(c, launch_minimizer_bin(c->stream, d_bases, d_offsets, n, P, threads, hist, c->d_state,
                                   c->d_min_slots, nullptr, nullptr, 0));
    if (!fits) return bin_long_reads(c, d_bases, d_offsets, n, P, hist);
    return HULK_OK;
}

// ---- RCCL, bound at run time.  libhulkhip.so does not carry a DT_NEEDED for librccl.so.1 (573 MB, half a second to map):
// a single-GPU host never loads it.  dlopen finds the copy a host process already holds (torch bundles one under the
// same SONAME) or the one next to the HIP runtime this library is linked against.
struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};
Rccl *rccl() {
    static Rccl R = [] {
        Rccl r;
        const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        if (const char *only = getenv("HULK_RCCL_LIB")) r.handle = dlopen(only, RTLD_NOW | RTLD_GLOBAL);      // this build and no other
        else for (const char *n : names) { r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.handle) break; }
        if (!r.handle) { const char *e = dlerror(); r.error = std::string("librccl.so.1 not found: ") + (e ? e : ""); return r; }
#define RCCL_SYM(f) do { r.f = (decltype(r.f))dlsym(r.handle, "nccl" #f); if (!r.f) r.error = "librccl lacks nccl" #f; } while (0)
        RCCL_SYM(GetUniqueId); RCCL_SYM(CommInitRank); RCCL_SYM(CommDestroy); RCCL_SYM(AllGather); RCCL_SYM(AllReduce);
        RCCL_SYM(GroupStart); RCCL_SYM(GroupEnd); RCCL_SYM(GetErrorString);
#undef RCCL_SYM
        return r;
    }();
    return &R;
}
int fail_nccl(hulk_ctx *c, ncclResult_t r, const char *what) {
    return fail(c, HULK_ERR_COMM, std::string(what) + ": " + (rccl()->GetErrorString ? rccl()->GetErrorString(r) : "RCCL error"));
}
#define NCCLCHK(c, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return fail_nccl((c), r_, #call); } while (0)

// host transport: the buffers cross through pinned memory and the caller's function moves them between the ranks
int comm_host_stage(hulk_ctx *c, size_t bytes) {
    if (bytes <= c->comm.h_stage_cap) return HULK_OK;
    if (c->comm.h_stage) hipHostFree(c->comm.h_stage);
    c->comm.h_stage = nullptr; c->comm.h_stage_cap = 0;
    HIPCHK(c, hipHostMalloc((void **)&c->comm.h_stage, bytes + bytes / 4, hipHostMallocDefault));
    c->comm.h_stage_cap = bytes + bytes / 4;
    return HULK_OK;
}
// all-gather of `bytes` per rank on stream s; in place when d_send == d_recv + rank * bytes
int comm_allgather(hulk_ctx *c, hipStream_t s, const void *d_send, void *d_recv, size_t bytes) {
    hulk_ctx::Comm &m = c->comm;
    if (bytes == 0) return HULK_OK;
    m.bytes_rx += (uint64_t)bytes * (m.world - 1);
    uint8_t *own = (uint8_t *)d_recv + (size_t)m.rank * bytes;
    switch (m.kind) {
        case 1: NCCLCHK(c, rccl()->AllGather(d_send, d_recv, bytes, ncclUint8, m.nccl, s)); return HULK_OK;
        case 2: {
            { const int rc = comm_host_stage(c, bytes * (m.world + 1)); if (rc != HULK_OK) return rc; }
            HIPCHK(c, hipMemcpyAsync(m.h_stage, d_send, bytes, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipStreamSynchronize(s));
            if (m.fn(m.user, HULK_XCHG_ALLGATHER, m.h_stage, m.h_stage + bytes, bytes) != 0)
                return fail(c, HULK_ERR_COMM, "the host's exchange function failed (all-gather)");
            HIPCHK(c, hipMemcpyAsync(d_recv, m.h_stage + bytes, bytes * m.world, hipMemcpyHostToDevice, s));
            HIPCHK(c, hipStreamSynchronize(s));
            return HULK_OK;
        }
        case 3:
            for (uint32_t r = 0; r < m.world; r++) {
                uint8_t *dst = (uint8_t *)d_recv + (size_t)r * bytes;
                if (dst != (const uint8_t *)d_send) HIPCHK(c, hipMemcpyAsync(dst, d_send, bytes, hipMemcpyDeviceToDevice, s));
            }
            return HULK_OK;
        default: break;
    }
    if ((const uint8_t *)d_send != own) HIPCHK(c, hipMemcpyAsync(own, d_send, bytes, hipMemcpyDeviceToDevice, s));
    return HULK_OK;
}
int comm_allreduce_u32(hulk_ctx *c, hipStream_t s, uint32_t *d_buf, size_t words) {
    hulk_ctx::Comm &m = c->comm;
    if (words == 0) return HULK_OK;
    m.bytes_rx += (uint64_t)words * 4 * 2 * (m.world - 1) / m.world;
    if (m.kind == 1) { NCCLCHK(c, rccl()->AllReduce(d_buf, d_buf, words, ncclUint32, ncclSum, m.nccl, s)); return HULK_OK; }
    if (m.kind == 2) {
        const size_t bytes = words * 4;
        { const int rc = comm_host_stage(c, bytes * 2); if (rc != HULK_OK) return rc; }
        HIPCHK(c, hipMemcpyAsync(m.h_stage, d_buf, bytes, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (m.fn(m.user, HULK_XCHG_ALLREDUCE_U32, m.h_stage, m.h_stage + bytes, bytes) != 0)
            return fail(c, HULK_ERR_COMM, "the host's exchange function failed (all-reduce)");
        HIPCHK(c, hipMemcpyAsync(d_buf, m.h_stage + bytes, bytes, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipStreamSynchronize(s));
    }
    return HULK_OK;                                             // loopback / no peers: the identity
}

// frees everything hulk_comm_init* set up (also after a failed ncclCommInitRank, so that the call can be repeated)
void comm_teardown(hulk_ctx *c) {
    hulk_ctx::Comm &m = c->comm;
    if (m.stream) hipStreamSynchronize(m.stream);                 // no collective in flight when the communicator goes
    if (m.nccl && rccl()->CommDestroy) rccl()->CommDestroy(m.nccl);
    hipFree(m.d_hdr); hipFree(m.d_delta); hipFree(m.d_gather); hipFree(m.d_sk);
    for (int i = 0; i < 2; i++) { if (m.h_hdr[i]) hipHostFree(m.h_hdr[i]); if (m.ev_hdr[i]) hipEventDestroy(m.ev_hdr[i]); }
    if (m.h_stage) hipHostFree(m.h_stage);
    if (m.ev_ready) hipEventDestroy(m.ev_ready);
    if (m.ev_done) hipEventDestroy(m.ev_done);
    if (m.stream) hipStreamDestroy(m.stream);
    m = hulk_ctx::Comm{};
}

// what stream s has queued so far -> the collectives' stream, and back
int comm_enter(hulk_ctx *c, hipStream_t s) {
    HIPCHK(c, hipEventRecord(c->comm.ev_ready, s));
    HIPCHK(c, hipStreamWaitEvent(c->comm.stream, c->comm.ev_ready, 0));
    return HULK_OK;
}
int comm_leave(hulk_ctx *c, hipStream_t s) {
    HIPCHK(c, hipEventRecord(c->comm.ev_done, c->comm.stream));
    HIPCHK(c, hipStreamWaitEvent(s, c->comm.ev_done, 0));
    return HULK_OK;
}

// The kernels of one flush: `fb.count` consecutive spectra of `hist` (starting at fb.ring_base) through count-min + CWS, on stream s.
int flush_kernels(hulk_ctx *c, hipStream_t s, uint32_t *hist, const FlushBatch &fb) {
    if (!c->scaling) HIPCHK(c, launch_count_used(s, hist, c->d_state, fb));    // (with decay k_elem_index delivers the count)
    {   // whole-batch bound on the counters as they stand BEFORE this batch is added (see k_flush_decide)
        HIPCHK(c, launch_flush_decide(s, c->d_ctr, c->cms_depth * c->cms_width, c->d_kminslot, c->d_weights, (int)c->slots,
                                      (int)c->slot_begin, c->d_state, fb, (c->prune && !c->drift && !c->no_skip && c->slots) ? 1 : 0));
    }
    if (c->scaling) {
        HIPCHK(c, launch_elem_index(s, hist, c->d_blkcnt, c->d_eidx, c->d_etot, fb, c->d_state));
        HIPCHK(c, launch_cmsd_binorder(s, hist, c->d_pos16, c->d_meta8, c->d_eidx, c->d_etot, c->d_ctrd, c->d_segadd,
                                       c->d_segfac, c->d_sege0, c->d_cstart, c->d_f64, c->d_rcp32, c->cms_depth,
                                       c->cms_width, c->row_stride, c->decay_weight, c->d_state, fb));
    } else {
        HIPCHK(c, launch_cms_binorder(s, hist, c->d_pos16, c->d_meta8, c->d_ctr, c->d_segsum, c->d_cbase, c->d_f64,
                                      c->d_rcp32, c->cms_depth, c->cms_width, c->row_stride, c->d_state, fb));
    }
    if (c->slots) {
        ProfileRec pr{};
        if ((c->profiling & 1)) {
            HIPCHK(c, hipEventCreateWithFlags(&pr.a, PROFILE_EVENT_FLAGS)); HIPCHK(c, hipEventCreateWithFlags(&pr.b, PROFILE_EVENT_FLAGS));
            HIPCHK(c, hipEventRecord(pr.a, s));
        }
        HIPCHK(c, launch_cws_scan(s, c->d_k32, c->d_rcp32, c->d_tilemin, (int)c->slots, c->ntiles,
                                  c->row_stride, c->d_state, fb, c->prune ? c->d_kmin32 : nullptr, c->d_rext,
                                  c->d_weights, (int)c->slot_begin, c->d_visited, c->drift ? c->decay_weight : 0.0, c->d_scanmap));
        c->scan_tiles_total += (uint64_t)((c->slots + SCAN_ROWS - 1) / SCAN_ROWS) * (uint64_t)c->ntiles * 4u;
        if ((c->profiling & 1)) { HIPCHK(c, hipEventRecord(pr.b, s)); c->prof.push_back(pr); }
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
bool no_overlap_mode() { static const bool v = getenv("HULK_NO_OVERLAP") != nullptr; return v; }
hipStream_t flush_stream_of(hulk_ctx *c) { return no_overlap_mode() ? c->stream : c->flush_stream; }

// queue the kernels of a prepared flush on the flush stream; `gate` (may be null): an event on the work stream they wait for
int issue_flush(hulk_ctx *c, hipEvent_t gate = nullptr) {
    if (!c->deferred.armed) return HULK_OK;
    c->deferred.armed = false;
    const FlushBatch fb = c->deferred.fb;
    const int ring = c->deferred.ring;
    hipStream_t s = flush_stream_of(c);
    // ev_binned orders the flush behind the binning; recorded on a caller's stream (hulk_flush_batch_after: the stream its
    // collective runs on) it has to be waited for even when the flush shares the work stream
    if (!no_overlap_mode() || c->deferred.use_dep) HIPCHK(c, hipStreamWaitEvent(s, c->ev_binned, 0));
    if (!no_overlap_mode() && gate) HIPCHK(c, hipStreamWaitEvent(s, gate, 0));
    uint32_t *hist = c->d_hist + (size_t)ring * (size_t)c->ring_n * (size_t)c->B;
    if (c->deferred.allreduce) {                                 // hulk_step_sliced: sum the ranks' spectra first
        int rc = comm_enter(c, s);
        if (rc == HULK_OK) rc = comm_allreduce_u32(c, c->comm.stream, hist + (size_t)fb.ring_base * (size_t)c->B, (size_t)fb.count * (size_t)c->B);
        if (rc == HULK_OK) rc = comm_leave(c, s);
        if (rc != HULK_OK) return rc;
    }
    { const int rc = flush_kernels(c, s, hist, fb); if (rc != HULK_OK) return rc; }
    HIPCHK(c, hipEventRecord(c->ev_flushed[ring], s));
    c->pending_flush[ring] = true;
    return HULK_OK;
}

// Flush `count` consecutive spectra of the ring (starting at ring_base) through count-min + CWS.
int flush_batch(hulk_ctx *c, uint32_t count, hipStream_t dep_stream = nullptr, bool use_dep = false, bool allreduce = false) {
    if (count == 0) return HULK_OK;
    int rc = ensure_tables(c);
    if (rc != HULK_OK) return rc;
    rc = issue_flush(c);                                        // (at most one flush is ever waiting)
    if (rc != HULK_OK) return rc;
    FlushBatch fb{};
    fb.ring_base = c->ring_base; fb.ring_n = c->ring_n; fb.count = count;
    fb.parity = (int)(c->flush_index & 1); fb.num_bins = c->B;
    // everything binned so far (or the caller's all-reduce on dep_stream) ends where this event is recorded
    HIPCHK(c, hipEventRecord(c->ev_binned, use_dep ? dep_stream : c->stream));
    c->deferred.armed = true; c->deferred.fb = fb; c->deferred.ring = c->cur_ring;
    c->deferred.use_dep = use_dep; c->deferred.allreduce = allreduce;
    c->flush_index++;
    // Queued at once.  (Holding the flush back until the NEXT batch's minimizer kernel had run — so that its LDS-heavy
    // count-min kernels would meet k_jump_bin, which needs no LDS, instead of k_minimizer_fast — was measured: C3-shaped
    // 8.9e8 vs 9.8e8 reads/s without the delay.  What does pay is that the next batch's minimizer and jump-hash kernels
    // no longer wait for this flush: only the histogram kernels behind them do, see bin_reads.)
    return issue_flush(c);
}

int do_flush(hulk_ctx *c) { return flush_batch(c, 1); }

int check_device_error(hulk_ctx *c) {
    { int rcs = sync_all(c); if (rcs != HULK_OK) return rcs; }
    DevState st{};
    HIPCHK(c, hipMemcpyAsync(&st, c->d_state, sizeof st, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (st.err != 0) { c->sticky = st.err; return fail(c, st.err); }
    return HULK_OK;
}

}  // namespace

extern "C" {

int hulk_abi_version(void) { return HULK_ABI_VERSION; }
#ifndef HULK_SOURCE_HASH
#define HULK_SOURCE_HASH "unknown"
#endif
#ifndef HULK_HIPCC_VERSION
#define HULK_HIPCC_VERSION "unknown"
#endif
const char *hulk_build_info(void) { return "abi=2 arch=gfx950 sources=" HULK_SOURCE_HASH " hipcc=" HULK_HIPCC_VERSION; }
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
    if (p.flags & ~(HULK_FLAG_GAMMA_CPYTHON | HULK_FLAG_NO_PRUNE | HULK_FLAG_NO_SKIP)) return fail(nullptr, HULK_ERR_ARG, "unknown flags");
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
    CHK_CREATE(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    const size_t B = (size_t)c->B, S = c->S, SL = c->slots;
    CHK_CREATE(dalloc(&c->d_state, 1));
    if (const char *e = getenv("HULK_BATCH")) { int v = atoi(e); if (v >= 1 && v <= SCAN_BATCH_MAX) c->T = (uint32_t)v; }
    c->ring_n = c->T + 1;
    const size_t T = c->T, RN = c->ring_n;
    CHK_CREATE(dalloc(&c->d_hist, 2 * RN * B));
    {   // the flush kernels fill the gaps the VALU-bound minimizer kernels leave: lowest queue priority
        // measured best (7.66e8 reads/s vs 7.59e8 default vs 7.37e8 highest) — the minimizer chain is the
        // critical path
        int lo = 0, hi = 0;
        CHK_CREATE(hipDeviceGetStreamPriorityRange(&lo, &hi));
        const char *pe = getenv("HULK_FLUSH_PRIORITY");
        const int prio = pe ? atoi(pe) : lo;   // `lo` = least priority (numerically greatest)
        CHK_CREATE(hipStreamCreateWithPriority(&c->flush_stream, hipStreamNonBlocking, prio));
    }
    CHK_CREATE(hipEventCreateWithFlags(&c->ev_binned, hipEventDisableTiming));
    CHK_CREATE(hipEventCreateWithFlags(&c->ev_flushed[0], hipEventDisableTiming));
    CHK_CREATE(hipEventCreateWithFlags(&c->ev_flushed[1], hipEventDisableTiming));
    CHK_CREATE(dalloc(&c->d_hist_tmp, B));
    CHK_CREATE(dalloc(&c->d_slow_count, 2));
    CHK_CREATE(hipMemsetAsync(c->d_slow_count, 0, 8, c->stream));
    CHK_CREATE(dalloc(&c->d_min_slots, (size_t)MIN_SLOTS));
    CHK_CREATE(hipMemsetAsync(c->d_min_slots, 0, (size_t)MIN_SLOTS * 8, c->stream));
    CHK_CREATE(dalloc(&c->d_ctr, (size_t)c->cms_depth * c->cms_width));
    CHK_CREATE(dalloc(&c->d_segsum, T * cms_binorder_entries(c->cms_depth, c->cms_width)));
    CHK_CREATE(dalloc(&c->d_cbase, T * cms_binorder_entries(c->cms_depth, c->cms_width)));
    CHK_CREATE(dalloc(&c->d_f64, T * B));
    CHK_CREATE(dalloc(&c->d_rcp32, T * c->row_stride));
    CHK_CREATE(dalloc(&c->d_mins, S));
    CHK_CREATE(dalloc(&c->d_weights, S));
    CHK_CREATE(dalloc(&c->d_rcb, SL * B * 3));
    CHK_CREATE(dalloc(&c->d_k32, SL * c->row_stride));
    CHK_CREATE(dalloc(&c->d_tilemin, T * ((SL + SCAN_ROWS - 1) / SCAN_ROWS) * SCAN_ROWS * (size_t)c->ntiles * 4));
    CHK_CREATE(dalloc(&c->d_kmin32, (SL ? SL : 1) * (size_t)c->ntiles * 4));
    CHK_CREATE(dalloc(&c->d_rext, T * (size_t)c->ntiles * 4 * 2));
    CHK_CREATE(dalloc(&c->d_kminslot, (SL ? SL : 1)));
    CHK_CREATE(dalloc(&c->d_scanmap, ((SL + SCAN_ROWS - 1) / SCAN_ROWS + 1) * (((size_t)c->ntiles * 4 + 63) / 64)));
    CHK_CREATE(dalloc(&c->d_slotmin, T * ((SL + SCAN_ROWS - 1) / SCAN_ROWS) * SCAN_ROWS));
    CHK_CREATE(dalloc(&c->d_visited, (size_t)MIN_SLOTS));
    CHK_CREATE(hipMemsetAsync(c->d_visited, 0, (size_t)MIN_SLOTS * 8, c->stream));
    // exact pruning of the K scan: without drift weights only fall; with drift (curMin = w / decayWeight) that still
    // holds for negative weights, which is what k_cws_scan tests then; decayRatio == 0 (decayWeight 0) is left alone
    c->prune = !(c->drift && c->decay_weight <= 0.0) && !getenv("HULK_NO_PRUNE") && !(p.flags & HULK_FLAG_NO_PRUNE);
    c->no_skip = getenv("HULK_NO_SKIP") != nullptr || (p.flags & HULK_FLAG_NO_SKIP) != 0;
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
    hipFree(c->d_state); hipFree(c->d_hist); hipFree(c->d_hist_tmp);
    hipFree(c->d_meta8); hipFree(c->d_segsum); hipFree(c->d_cbase);
    hipFree(c->d_segadd); hipFree(c->d_segfac); hipFree(c->d_cstart); hipFree(c->d_sege0);
    hipFree(c->d_ctr); hipFree(c->d_pos16); hipFree(c->d_mins); hipFree(c->d_f64); hipFree(c->d_weights);
    hipFree(c->d_blkcnt); hipFree(c->d_eidx); hipFree(c->d_etot); hipFree(c->d_ctrd);
    hipFree(c->d_candA); hipFree(c->d_candB); hipFree(c->d_rcb); hipFree(c->d_rcp32); hipFree(c->d_k32); hipFree(c->d_tilemin);
    hipFree(c->d_scanmap); hipFree(c->d_slotmin); hipFree(c->d_kmin32); hipFree(c->d_rext); hipFree(c->d_visited); hipFree(c->d_kminslot);
    for (auto &hs : c->hstage) {
        if (hs.ev) hipEventDestroy(hs.ev);
        if (hs.h_bases) hipHostFree(hs.h_bases);
        if (hs.h_off) hipHostFree(hs.h_off);
        hipFree(hs.d_bases); hipFree(hs.d_off);
    }
    hipFree(c->d_min_slots); hipFree(c->d_slow_list); hipFree(c->d_slow_count);
    hipFree(c->ml.x); hipFree(c->ml.slot); hipFree(c->ml.key); hipFree(c->ml.cnt); hipFree(c->ml.off); hipFree(c->ml.bsum); hipFree(c->ml.partial); hipFree(c->ml.nib); hipFree(c->ml.nib_over); hipFree(c->ml.lo); hipFree(c->ml.lo_cnt); hipFree(c->ml.dmask); hipFree(c->ml.dsum);
    hipFree(c->d_long_xs); hipFree(c->d_long_valid); hipFree(c->d_long_table); hipFree(c->d_long_desc);
    comm_teardown(c);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
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

// internal accessors for hulk_ingest.hip (not part of the ABI)
namespace hulk {
hipStream_t ctx_stream(hulk_ctx *c) { return c->stream; }
uint64_t ctx_min_read_len(const hulk_ctx *c) { return (uint64_t)c->p.w + c->p.k - 1; }
int ctx_fail(hulk_ctx *c, int code, const char *full_message) {
    c->last_error = (full_message && *full_message) ? full_message : err_text(code);
    return code;
}
}  // namespace hulk


namespace {
// a staging set that is free again (its last copies and kernels done) and holds nbytes of bases and cn reads
int stage_ready(hulk_ctx *c, hulk_ctx::HostStage &hs, size_t nbytes, uint64_t cn) {
    if (!hs.ev) HIPCHK(c, hipEventCreateWithFlags(&hs.ev, hipEventDisableTiming));
    if (hs.busy) { HIPCHK(c, hipEventSynchronize(hs.ev)); hs.busy = false; }     // its copies and kernels are done
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
        static const unsigned tmax = [] { const char *e = getenv("HULK_HOST_COPY_THREADS"); const long v = e ? atol(e) : 4; return (unsigned)(v < 1 ? 1 : v > 32 ? 32 : v); }();
        const unsigned T = nbytes >= (8u << 20) ? tmax : 1u;
        const size_t piece = (nbytes / T + 63) & ~(size_t)63;
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
}  // namespace

namespace hulk {
int ctx_stage_acquire(hulk_ctx *c, size_t nbytes, uint64_t n, StageSet *out) {
    hulk_ctx::HostStage &hs = c->hstage[c->hstage_cur];
    const int rc = stage_ready(c, hs, nbytes, n);
    if (rc != HULK_OK) return rc;
    out->h_bases = hs.h_bases; out->d_bases = hs.d_bases; out->h_off = hs.h_off; out->d_off = hs.d_off; out->cap_bases = hs.cap_bases;
    return HULK_OK;
}
int ctx_stage_release(hulk_ctx *c) {
    hulk_ctx::HostStage &hs = c->hstage[c->hstage_cur];
    HIPCHK(c, hipEventRecord(hs.ev, c->stream));
    hs.busy = true;
    c->hstage_cur ^= 1;
    return HULK_OK;
}
}  // namespace hulk

extern "C" {
int hulk_add_reads_device(hulk_ctx *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n,
                          uint32_t max_read_len, uint64_t bases_bytes) {
    if (!c) return HULK_ERR_ARG;
    if (c->finished) return fail(c, HULK_ERR_STATE, "context already finished");
    if (c->sticky != HULK_OK) return fail(c, c->sticky);
    if (n && (!d_bases || !d_offsets)) return fail(c, HULK_ERR_ARG, "NULL buffer");
    const uint64_t I = c->p.interval;
    uint64_t pos = 0;
    while (pos < n) {
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
    if (c->finished) return fail(c, HULK_ERR_STATE, "context already finished");
    if (n && (!d_bases || !d_offsets)) return fail(c, HULK_ERR_ARG, "NULL buffer");
    if (c->ring_base != 0) return fail(c, HULK_ERR_STATE, "a partial interval is pending");
    if (first_spectrum && !reads_per_spectrum) return fail(c, HULK_ERR_ARG, "first_spectrum needs reads_per_spectrum");
    const uint64_t count = reads_per_spectrum ? (n + reads_per_spectrum - 1) / reads_per_spectrum : (n ? 1u : 0u);
    if ((uint64_t)first_spectrum + count > c->T) return fail(c, HULK_ERR_ARG, "more spectra than the batch size");
    const uint64_t skip = (uint64_t)first_spectrum * reads_per_spectrum;       // as if that many reads had been binned before
    for (uint64_t pos = 0; pos < n; pos += MAX_READS_PER_LAUNCH) {
        const uint64_t chunk = std::min<uint64_t>(MAX_READS_PER_LAUNCH, n - pos);
        int rc = bin_reads(c, d_bases, d_offsets + pos, chunk, max_read_len, bases_bytes, reads_per_spectrum, skip + pos);
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
        const int rc = hulk_add_reads_device(c, hs.d_bases, hs.d_off, cn, (uint32_t)cmax, hs.cap_bases);
        if (rc != HULK_OK) return rc;
        HIPCHK(c, hipEventRecord(hs.ev, c->stream));
        hs.busy = true;
        i0 = i1;
    }
    return HULK_OK;
}

int hulk_add_histogram(hulk_ctx *c, const uint32_t *bins) {
    if (!c || !bins) return fail(c, HULK_ERR_ARG, "NULL");
    HIPCHK(c, hipMemcpyAsync(c->d_hist_tmp, bins, (size_t)c->B * 4, hipMemcpyHostToDevice, c->stream));
    { int rcw = ring_ready_for_writes(c); if (rcw != HULK_OK) return rcw; }
    HIPCHK(c, launch_add_hist(c->stream, ring_hist(c) + (size_t)c->ring_base * (size_t)c->B, c->d_hist_tmp, c->B));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->hist_hook_used = true;
    return HULK_OK;
}

int hulk_flush(hulk_ctx *c) {
    if (!c) return HULK_ERR_ARG;
    if (c->finished) return fail(c, HULK_ERR_STATE, "context already finished");
    // after hulk_bin_reads_device filled several spectra a flush of ONE would leave the others behind for hulk_finish to
    // flush a second time: they go through hulk_flush_batch[_after]
    if (c->bin_spectra > 1) return fail(c, HULK_ERR_STATE, "hulk_bin_reads_device filled several spectra: use hulk_flush_batch");
    const int rc = do_flush(c);
    if (rc == HULK_OK) c->bin_spectra = 0;
    return rc;
}

int hulk_finish(hulk_ctx *c) {
    if (!c) return HULK_ERR_ARG;
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

int hulk_smash(int device, const uint64_t *mins, const double *weights, uint32_t n_sketches, uint32_t sketch_size,
               int metric, double *distances) {
    if (!mins || !weights || !distances) return fail(nullptr, HULK_ERR_ARG, "NULL");
    if (metric != HULK_METRIC_JACCARD && metric != HULK_METRIC_WEIGHTED_JACCARD) return fail(nullptr, HULK_ERR_ARG, "metric");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, HULK_ERR_NO_DEVICE);
    if (device < 0 || device >= ndev) return fail(nullptr, HULK_ERR_ARG, "device ordinal");
#define SM_CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { hipFree(d_m); hipFree(d_w); hipFree(d_o); return fail_hip(nullptr, e_, #call); } } while (0)
    unsigned long long *d_m = nullptr; double *d_w = nullptr, *d_o = nullptr;
    const size_t NS = (size_t)n_sketches * sketch_size, NN = (size_t)n_sketches * n_sketches;
    SM_CHK(hipSetDevice(device));
    SM_CHK(hipMalloc((void **)&d_m, (NS ? NS : 1) * 8));
    SM_CHK(hipMalloc((void **)&d_w, (NS ? NS : 1) * 8));
    SM_CHK(hipMalloc((void **)&d_o, (NN ? NN : 1) * 8));
    SM_CHK(hipMemcpy(d_m, mins, NS * 8, hipMemcpyHostToDevice));
    SM_CHK(hipMemcpy(d_w, weights, NS * 8, hipMemcpyHostToDevice));
    SM_CHK(launch_smash(nullptr, d_m, d_w, n_sketches, sketch_size, metric, d_o));
    SM_CHK(hipMemcpy(distances, d_o, NN * 8, hipMemcpyDeviceToHost));
#undef SM_CHK
    hipFree(d_m); hipFree(d_w); hipFree(d_o);
    return HULK_OK;
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

int hulk_synchronize(hulk_ctx *c) {
    if (!c) return HULK_ERR_ARG;
    return sync_all(c);
}

int hulk_set_profiling(hulk_ctx *c, int enabled) {
    if (!c) return HULK_ERR_ARG;
    // 1 (the on/off switch) = every instrumented kernel; otherwise a mask: 2 k_minimizer_fast, 4 k_jump_bin, 8 k_cws_scan
    c->profiling = enabled == 1 ? 7 : ((enabled & 6) | ((enabled & 8) ? 1 : 0));
    return HULK_OK;
}

int hulk_get_profile(hulk_ctx *c, const char *kernel, uint64_t *launches, double *total_ms) {
    if (!c || !launches || !total_ms) return fail(c, HULK_ERR_ARG, "NULL");
    int which = 0;
    if (kernel && strcmp(kernel, "k_minimizer_fast") == 0) which = 1;
    else if (kernel && strcmp(kernel, "k_jump_bin") == 0) which = 2;
    else if (kernel && strcmp(kernel, "k_jump_left") == 0) which = 3;
    else if (kernel && strcmp(kernel, "k_cws_scan") != 0)
        return fail(c, HULK_ERR_ARG, "instrumented kernels: k_cws_scan, k_minimizer_fast, k_jump_bin, k_jump_left");
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
// ------------------------------------------------------------------------------------------------------------------
namespace {
int comm_setup(hulk_ctx *c, int kind, uint32_t rank, uint32_t world) {
    hulk_ctx::Comm &m = c->comm;
    if (m.kind != 0) return fail(c, HULK_ERR_STATE, "the context already has a communicator");
    if (world == 0 || rank >= world) return fail(c, HULK_ERR_ARG, "rank / world");
    if (c->seq_count || c->flush_index) return fail(c, HULK_ERR_STATE, "hulk_comm_init must precede the first read");
    HIPCHK(c, hipSetDevice(c->p.device));
    const size_t NC = (size_t)c->cms_depth * c->cms_width;
    HIPCHK(c, dalloc(&m.d_hdr, (size_t)world * SHARD_HDR));
    HIPCHK(c, dalloc(&m.d_delta, (size_t)world * c->T * NC));
    HIPCHK(c, dalloc(&m.d_sk, (size_t)world * (2 + 2 * (size_t)c->S)));
    HIPCHK(c, hipMemset(m.d_hdr, 0, (size_t)world * SHARD_HDR * 4));
    {
        int lo = 0, hi = 0;
        HIPCHK(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(c, hipStreamCreateWithPriority(&m.stream, hipStreamNonBlocking, hi));   // `hi` = greatest priority
        HIPCHK(c, hipEventCreateWithFlags(&m.ev_ready, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&m.ev_done, hipEventDisableTiming));
    }
    for (int i = 0; i < 2; i++) {
        HIPCHK(c, hipHostMalloc((void **)&m.h_hdr[i], (size_t)world * SHARD_HDR * 4, hipHostMallocDefault));
        HIPCHK(c, hipEventCreateWithFlags(&m.ev_hdr[i], hipEventDisableTiming));
    }
    m.rank = rank; m.world = world; m.kind = kind;
    return HULK_OK;
}
// intervals of a step that rank r holds (hulk_hip.h: whole intervals, T per rank, in rank order)
uint32_t shard_count(const hulk_ctx *c, uint32_t step_intervals, uint32_t r) {
    const uint64_t lo = (uint64_t)r * c->T;
    if (step_intervals <= lo) return 0;
    return (uint32_t)std::min<uint64_t>(c->T, step_intervals - lo);
}
}  // namespace

int hulk_comm_unique_id(void *unique_id) {
    if (!unique_id) return fail(nullptr, HULK_ERR_ARG, "NULL");
    Rccl *R = rccl();
    if (!R->error.empty()) return fail(nullptr, HULK_ERR_COMM, R->error);
    static_assert(sizeof(ncclUniqueId) == HULK_UNIQUE_ID_BYTES, "ncclUniqueId size");
    NCCLCHK(nullptr, R->GetUniqueId((ncclUniqueId *)unique_id));
    return HULK_OK;
}

int hulk_comm_init(hulk_ctx *c, const void *unique_id, uint32_t rank, uint32_t world) {
    if (!c || !unique_id) return fail(c, HULK_ERR_ARG, "NULL");
    Rccl *R = rccl();
    if (!R->error.empty()) return fail(c, HULK_ERR_COMM, R->error);
    { const int rc = comm_setup(c, 1, rank, world); if (rc != HULK_OK) return rc; }
    ncclUniqueId id; memcpy(&id, unique_id, sizeof id);
    const ncclResult_t r = R->CommInitRank(&c->comm.nccl, (int)world, id, (int)rank);
    if (r != ncclSuccess) { const int rc = fail_nccl(c, r, "ncclCommInitRank"); c->comm.nccl = nullptr; comm_teardown(c); return rc; }
    return HULK_OK;
}

int hulk_comm_init_host(hulk_ctx *c, uint32_t rank, uint32_t world, hulk_exchange_fn fn, void *user) {
    if (!c || !fn) return fail(c, HULK_ERR_ARG, "NULL");
    { const int rc = comm_setup(c, 2, rank, world); if (rc != HULK_OK) return rc; }
    c->comm.fn = fn; c->comm.user = user;
    return HULK_OK;
}

int hulk_comm_init_loopback(hulk_ctx *c, uint32_t rank, uint32_t world) {
    if (!c) return HULK_ERR_ARG;
    return comm_setup(c, 3, rank, world);
}

int hulk_step_sharded(hulk_ctx *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n, uint32_t max_read_len,
                      uint64_t bases_bytes, uint32_t step_intervals) {
    if (!c) return HULK_ERR_ARG;
    hulk_ctx::Comm &m = c->comm;
    if (m.kind == 0) return fail(c, HULK_ERR_STATE, "hulk_step_sharded needs hulk_comm_init");
    if (c->finished) return fail(c, HULK_ERR_STATE, "context already finished");
    if (c->sticky != HULK_OK) return fail(c, c->sticky);
    const uint64_t I = c->p.interval;
    if (I == 0) return fail(c, HULK_ERR_ARG, "hulk_step_sharded needs params.interval > 0 (the global sketching interval)");
    if (c->ring_base != 0 || c->bin_spectra) return fail(c, HULK_ERR_STATE, "a partial interval / an unflushed batch is pending");
    if (step_intervals == 0 || step_intervals > (uint64_t)m.world * c->T) return fail(c, HULK_ERR_ARG, "step_intervals");
    const uint32_t own = shard_count(c, step_intervals, m.rank);
    if ((own == 0) != (n == 0) || n > (uint64_t)own * I || (own && n <= (uint64_t)(own - 1) * I))
        return fail(c, HULK_ERR_ARG, "n_reads does not match this rank's intervals of the step");
    if (n && (!d_bases || !d_offsets)) return fail(c, HULK_ERR_ARG, "NULL buffer");
    int rc = ensure_tables(c);
    if (rc != HULK_OK) return rc;
    // 1. bin this rank's intervals into spectra 0 .. own-1 of the current ring (work stream)
    for (uint64_t pos = 0; pos < n; pos += MAX_READS_PER_LAUNCH) {
        const uint64_t chunk = std::min<uint64_t>(MAX_READS_PER_LAUNCH, n - pos);
        rc = bin_reads(c, d_bases, d_offsets + pos, chunk, max_read_len, bases_bytes, I, pos);
        if (rc != HULK_OK) return rc;
    }
    c->seq_count += n;
    rc = issue_flush(c);
    if (rc != HULK_OK) return rc;
    // 2. which exchange: the verdicts of the step before (they travelled with its exchange) — any rank's need_full keeps
    //    the spectra exchange.  The wait ends when the previous step's exchange has run: this step's binning is queued.
    bool full = c->drift || c->scaling || !c->prune || c->no_skip || m.step == 0;
    if (!full) {
        const int prev = (int)((m.step - 1) & 1);
        if (m.hdr_pending[prev]) { HIPCHK(c, hipEventSynchronize(m.ev_hdr[prev])); m.hdr_pending[prev] = false; }
        for (uint32_t r = 0; r < m.world; r++) if (m.h_hdr[prev][(size_t)r * SHARD_HDR + 1]) full = true;
    }
    static const bool force_full = getenv("HULK_SHARD_FULL") != nullptr;      // A/B aid: always the spectra exchange
    if (force_full) full = true;
    hipStream_t s = flush_stream_of(c);
    HIPCHK(c, hipEventRecord(c->ev_binned, c->stream));
    if (!no_overlap_mode()) HIPCHK(c, hipStreamWaitEvent(s, c->ev_binned, 0));
    const int ring = c->cur_ring;
    uint32_t *hist = ring_hist(c);
    const size_t B = (size_t)c->B, NC = (size_t)c->cms_depth * c->cms_width;
    uint32_t *own_hdr = m.d_hdr + (size_t)m.rank * SHARD_HDR;
    FlushBatch fb{};
    fb.ring_base = 0; fb.ring_n = c->ring_n; fb.count = own; fb.parity = 0; fb.num_bins = c->B;
    HIPCHK(c, hipMemsetAsync(own_hdr, 0, SHARD_HDR * 4, s));
    // this rank's verdict for the NEXT step: the whole-batch bound on the counters and weights as they stand now
    // (a rank without slots has nothing to protect: its verdict stays 0)
    if (c->slots)
        HIPCHK(c, launch_flush_decide(s, c->d_ctr, (int)NC, c->d_kminslot, c->d_weights, (int)c->slots, (int)c->slot_begin,
                                      c->d_state, fb, 1, own_hdr + 1));
    if (!full) {
        uint32_t *own_delta = m.d_delta + (size_t)m.rank * c->T * NC;
        HIPCHK(c, hipMemsetAsync(own_delta, 0, (size_t)c->T * NC * 4, s));
        HIPCHK(c, launch_shard_local(s, hist, c->d_pos16, own_hdr, own_delta, c->cms_depth, c->cms_width, fb));
        HIPCHK(c, hipEventRecord(c->ev_flushed[ring], s));          // the ring is wiped: the work stream may fill it again
        c->pending_flush[ring] = true;
        rc = comm_enter(c, s);
        if (rc != HULK_OK) return rc;
        if (m.kind == 1) NCCLCHK(c, rccl()->GroupStart());
        rc = comm_allgather(c, m.stream, own_hdr, m.d_hdr, SHARD_HDR * 4);
        const int rc2 = rc == HULK_OK ? comm_allgather(c, m.stream, own_delta, m.d_delta, (size_t)c->T * NC * 4) : rc;
        if (m.kind == 1) NCCLCHK(c, rccl()->GroupEnd());                 // (closed whatever the calls inside it returned)
        if (rc2 != HULK_OK) return rc2;
        rc = comm_leave(c, s);
        if (rc != HULK_OK) return rc;
        HIPCHK(c, launch_shard_apply(s, m.d_hdr, m.d_delta, c->d_ctr, c->cms_depth, c->cms_width, m.world, c->T,
                                     step_intervals, c->B, c->d_state));
        m.steps_delta++;
    } else {
        const size_t need = (size_t)m.world * c->T * B;
        if (need > m.gather_words) {
            HIPCHK(c, hipStreamSynchronize(s));
            hipFree(m.d_gather); m.d_gather = nullptr; m.gather_words = 0;
            HIPCHK(c, hipMalloc((void **)&m.d_gather, need * 4));
            m.gather_words = need;
        }
        rc = comm_enter(c, s);
        if (rc == HULK_OK) rc = comm_allgather(c, m.stream, own_hdr, m.d_hdr, SHARD_HDR * 4);
        if (rc == HULK_OK) rc = comm_allgather(c, m.stream, hist, m.d_gather, (size_t)c->T * B * 4);
        if (rc == HULK_OK) rc = comm_leave(c, s);
        if (rc != HULK_OK) return rc;
        if (own) HIPCHK(c, hipMemsetAsync(hist, 0, (size_t)own * B * 4, s));    // Wipe of the rank's own copy
        HIPCHK(c, hipEventRecord(c->ev_flushed[ring], s));
        c->pending_flush[ring] = true;
        for (uint32_t r = 0; r < m.world; r++) {                    // the ordinary flush of every rank's intervals, stream order
            const uint32_t cnt = shard_count(c, step_intervals, r);
            if (!cnt) break;
            FlushBatch fr{};
            fr.ring_base = 0; fr.ring_n = c->T; fr.count = cnt; fr.parity = (int)(c->flush_index & 1); fr.num_bins = c->B;
            c->flush_index++;
            rc = flush_kernels(c, s, m.d_gather + (size_t)r * c->T * B, fr);
            if (rc != HULK_OK) return rc;
        }
        m.steps_full++;
    }
    const int cur = (int)(m.step & 1);
    HIPCHK(c, hipMemcpyAsync(m.h_hdr[cur], m.d_hdr, (size_t)m.world * SHARD_HDR * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipEventRecord(m.ev_hdr[cur], s));
    m.hdr_pending[cur] = true;
    m.step++;
    m.global_intervals += step_intervals;
    c->cur_ring ^= 1;
    return HULK_OK;
}

int hulk_step_sharded_host(hulk_ctx *c, const uint8_t *bases, const uint64_t *offsets, uint64_t n, uint32_t step_intervals) {
    if (!c) return HULK_ERR_ARG;
    if (n && (!bases || !offsets)) return fail(c, HULK_ERR_ARG, "NULL buffer");
    if (n == 0) return hulk_step_sharded(c, nullptr, nullptr, 0, 0, 0, step_intervals);
    uint64_t max_len = 0;
    { const int rcv = check_host_reads(c, offsets, n, &max_len); if (rcv != HULK_OK) return rcv; }
    hulk_ctx::HostStage *hs = nullptr;
    { const int rcs = stage_host_reads(c, bases, offsets, 0, n, &hs); if (rcs != HULK_OK) return rcs; }
    const int rc = hulk_step_sharded(c, hs->d_bases, hs->d_off, n, (uint32_t)max_len, hs->cap_bases, step_intervals);
    HIPCHK(c, hipEventRecord(hs->ev, c->stream));               // (the binning kernels are on the work stream)
    hs->busy = true;
    return rc;
}

int hulk_step_sliced(hulk_ctx *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n, uint32_t max_read_len,
                     uint64_t bases_bytes, uint64_t reads_per_spectrum, uint32_t n_spectra) {
    if (!c) return HULK_ERR_ARG;
    if (c->comm.kind == 0) return fail(c, HULK_ERR_STATE, "hulk_step_sliced needs hulk_comm_init");
    if (n_spectra == 0 || n_spectra > c->T) return fail(c, HULK_ERR_ARG, "n_spectra");
    int rc = hulk_bin_reads_device_at(c, d_bases, d_offsets, n, max_read_len, bases_bytes, reads_per_spectrum, 0);
    if (rc != HULK_OK) return rc;
    if (c->bin_spectra > n_spectra) return fail(c, HULK_ERR_ARG, "more spectra binned than n_spectra");
    rc = flush_batch(c, n_spectra, nullptr, false, true);
    if (rc == HULK_OK) { c->cur_ring ^= 1; c->bin_spectra = 0; }
    return rc;
}

int hulk_gather_sketch(hulk_ctx *c, uint64_t *mins, double *weights) {
    if (!c || !mins || !weights) return fail(c, HULK_ERR_ARG, "NULL");
    hulk_ctx::Comm &m = c->comm;
    if (m.kind == 0 || m.world == 1) return hulk_get_sketch(c, mins, weights);
    { int rcs = sync_all(c); if (rcs != HULK_OK) return rcs; }
    const size_t S = c->S, blk = 2 + 2 * S;
    std::vector<unsigned long long> h((size_t)m.world * blk);
    unsigned long long *own = m.d_sk + (size_t)m.rank * blk;
    const unsigned long long head[2] = {c->slot_begin, c->slots};
    HIPCHK(c, hipMemcpyAsync(own, head, 16, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(own + 2, c->d_mins, S * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(own + 2 + S, c->d_weights, S * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));                     // `head` is a stack buffer
    { const int rc = comm_allgather(c, c->stream, own, m.d_sk, blk * 8); if (rc != HULK_OK) return rc; }
    HIPCHK(c, hipMemcpyAsync(h.data(), m.d_sk, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < S; i++) { mins[i] = 0; weights[i] = 1.7976931348623157e308; }
    for (uint32_t r = 0; r < m.world; r++) {
        const unsigned long long *b = h.data() + (size_t)r * blk;
        const uint64_t sb = b[0], sc = b[1];
        if (sb + sc > S) return fail(c, HULK_ERR_COMM, "a rank reported a slot shard outside the sketch");
        for (uint64_t i = sb; i < sb + sc; i++) { mins[i] = b[2 + i]; memcpy(&weights[i], &b[2 + S + i], 8); }
    }
    return HULK_OK;
}

int hulk_get_comm_stats(hulk_ctx *c, uint64_t *steps_delta, uint64_t *steps_full, uint64_t *bytes_received) {
    if (!c) return HULK_ERR_ARG;
    if (steps_delta) *steps_delta = c->comm.steps_delta;
    if (steps_full) *steps_full = c->comm.steps_full;
    if (bytes_received) *bytes_received = c->comm.bytes_rx;
    return HULK_OK;
}

}  // extern "C"
