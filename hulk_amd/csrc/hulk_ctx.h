// hulk_ctx.h — the context behind the C ABI and the host-side pieces that share it (private to libhulkhip.so):
//   hulk_api.hip     the ABI's entry points: create / destroy, AddSeq variants, getters, profiling, smash
//   hulk_tables.hip  count-min chain tables and the CWS parameter tables (newCWS, histosketch.go:95-126)
//   hulk_flush.hip   orchestration of a batch: binning launches on the work stream(s), the spectrum rings, flushes on the
//                    flush stream, host staging
//   hulk_comm.hip    multi-GPU: RCCL binding, host / loopback transports, hulk_step_sharded / hulk_step_sliced, gather
// Host code only: every numeric step of the path runs in the kernels of hulk_minimizer / hulk_spectrum / hulk_countmin /
// hulk_cws .hip; there is no CPU fallback.
#pragma once
#include "../../include/hulk_hip.h"
#include "hulk_internal.h"

#include <rccl/rccl.h>      // types and prototypes only: librccl.so.1 is bound at run time (hulk_comm_init), see hulk_comm.hip

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// Debug aid: HULK_POISON=<byte> fills every device / pinned allocation of this file with that byte, so that a read of
// memory nothing has written shows up the same way in every process (tools/fuzz_parity.py found one such read by its
// dependence on what earlier contexts had left behind).
static inline int poison_byte() {
    static const int v = [] { const char *e = HULK_EXP_ENV("HULK_POISON"); return e ? (int)(strtol(e, nullptr, 0) & 0xff) : -1; }();
    return v;
}
static inline hipError_t poison_malloc(void **p, size_t n) {
    hipError_t e = (hipMalloc)(p, n);
    if (e == hipSuccess && poison_byte() >= 0 && n) { e = hipMemset(*p, poison_byte(), n); if (e == hipSuccess) e = hipDeviceSynchronize(); }
    return e;
}
static inline hipError_t poison_host_malloc(void **p, size_t n, unsigned flags) {
    hipError_t e = (hipHostMalloc)(p, n, flags);
    if (e == hipSuccess && poison_byte() >= 0 && n) memset(*p, poison_byte(), n);
    return e;
}
#define hipMalloc(p, n) poison_malloc((void **)(p), (n))
#define hipHostMalloc(p, n, f) poison_host_malloc((void **)(p), (n), (f))

namespace hulk {
constexpr uint64_t MAX_READS_PER_LAUNCH = 4u << 20;   // 4 Mi reads -> <= ~7 GB of minimizer list at w = 9
struct ProfMark { hipEvent_t e; const char *kernel; hipStream_t s; };   // hulk_set_profiling bit 32 (prof_mark)
struct ProfileRec { hipEvent_t a, b; int which; };   // which: 0 = k_cws_scan, 1 = k_minimizer_fast, 2 = k_jump_bin, 3 = k_jump_left, 4 = k_cmsd_freq
}  // namespace hulk

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
    struct PreparedFlush { bool armed = false; hulk::FlushBatch fb{}; int ring = 0;
                           bool use_dep = false;        // ev_binned was recorded on a caller's stream (hulk_flush_batch_after)
                           bool allreduce = false;      // hulk_step_sliced: the spectra are summed over the ranks first
    } deferred;   // a flush between its preparation (ev_binned recorded) and the queueing of its kernels
    // the exchange of a multi-rank run (hulk_comm_init*, hulk_step_sharded / hulk_step_sliced)
    struct Comm {
        int kind = 0;                                   // 0 none, 1 RCCL, 2 host callback, 3 loopback
        uint32_t rank = 0, world = 1;
        ncclComm_t nccl = nullptr;
        hulk_exchange_fn fn = nullptr; void *user = nullptr;
        // (the collectives run on the flush stream, in line with the kernels around them: hulk_comm.hip, comm_exchange)
        uint32_t *d_hdr = nullptr;                      // [world][SHARD_HDR]: {need_full, step + 1 (the seal), used bins per interval ...}
        uint32_t *d_delta = nullptr;                    // [world][T][depth * width] count-min increments per interval
        uint32_t *d_gather = nullptr; size_t gather_words = 0;   // [world][T][num_bins] spectra of a full exchange
        uint32_t *h_hdr[2] = {nullptr, nullptr}; hipEvent_t ev_hdr[2] = {nullptr, nullptr}; bool hdr_pending[2] = {false, false};
        uint64_t hdr_resyncs = 0;                       // times a header was not where its copy / event said it was, and was fetched again
        uint64_t hdr_void = 0;                          // times a rank's block of the previous step carried another step's seal
        uint32_t inject = 0; uint64_t inject_step = 0;  // hulk_debug_inject (test hook)
        uint8_t *h_stage = nullptr; size_t h_stage_cap = 0;      // host transport: pinned staging
        unsigned long long *d_sk = nullptr;             // hulk_gather_sketch: [world][2 + 2 S]
        uint64_t step = 0, steps_delta = 0, steps_full = 0, bytes_rx = 0;
        uint64_t global_intervals = 0;                  // intervals of the GLOBAL stream the steps so far covered
    } comm;
    // device state
    hulk::DevState *d_state = nullptr;
    uint32_t *d_hist = nullptr, *d_hist_tmp = nullptr;
    unsigned long long *d_ctr = nullptr, *d_mins = nullptr, *d_min_slots = nullptr;
    uint16_t *d_pos16 = nullptr;
    uint8_t *d_meta8 = nullptr; uint32_t *d_segsum = nullptr; unsigned long long *d_cbase = nullptr;   // bin-order count-min
    double *d_segadd = nullptr, *d_segfac = nullptr, *d_cstart = nullptr; uint32_t *d_sege0 = nullptr; // ... with decay
    double *d_f64 = nullptr, *d_weights = nullptr, *d_rcb = nullptr;
    float *d_rcp32 = nullptr, *d_k32 = nullptr, *d_tilemin = nullptr;
    float *d_rmm = nullptr;                                                        // [2][row_stride]: k_rcp_minmax (per-bin max / min of the batch's reciprocal vectors)
    unsigned long long *d_scanmap = nullptr;                                       // [slot groups][wave tiles / 64]: k_scan_test's verdicts
    uint32_t *d_scanlist = nullptr, *d_scanlist_n = nullptr;                       // ... and as a list (slot group << 12 | wave tile) + its length
    float *d_slotmin = nullptr;                                                    // [T][slot groups][8]: k_slot_tmin (concept drift only)
    float *d_kmin32 = nullptr, *d_rext = nullptr, *d_kminslot = nullptr;          // bound test of k_cws_scan (no concept drift only)
    unsigned long long *d_visited = nullptr; uint64_t scan_tiles_total = 0; bool prune = false, no_skip = false;
    double *d_candA = nullptr; int32_t *d_candB = nullptr;
    // staging for host reads
    // hulk_add_reads (host buffers): two sets of pinned + device staging; the copy of chunk i+1 into pinned memory
    // and over PCIe runs while the kernels of chunk i do
    struct HostStage {
        uint8_t *h_bases = nullptr, *d_bases = nullptr; uint64_t *h_off = nullptr, *d_off = nullptr;
        size_t cap_bases = 0, cap_off = 0; hipEvent_t ev = nullptr, ev1 = nullptr; bool busy = false, busy1 = false;
    } hstage[2];
    int hstage_cur = 0;
    // Work lanes (hulk_flush.hip, lane_stream): every launch that touches spectrum ring r runs on lane r's stream — lane 0 the
    // context's stream, lane 1 a stream of its own, created with the first batch of ring 1 — so consecutive batches overlap.
    // A lane owns the per-launch buffers of the short-read kernels: the minimizer list (grow-only) and the list of reads the
    // fast kernel deferred (built by k_region_offsets)
    struct BinLane {
        hipStream_t stream = nullptr;                   // lane 1 only (lane 0 runs on the context's stream)
        hulk::MinimizerList ml{}; uint64_t ml_regions = 0;
        uint32_t *d_slow_list = nullptr, *d_slow_count = nullptr; uint64_t d_slow_cap = 0;
    } lane[2];
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;    // context stream -> lane 1 in front of a batch, and back (lanes_join)
    hipStream_t last_bin_stream = nullptr;              // the stream the latest binning launches went to
    hipEvent_t ev_heavy[2] = {nullptr, nullptr}; int heavy_idx = 0; bool heavy_set[2] = {false, false};   // experiment HULK_C3_GATE (profiling build)
    hipEvent_t ev_hold = nullptr;                                                                         // experiment HULK_C3_HOLD (profiling build)
    int stagger = 1; hipEvent_t ev_stagger = nullptr;   // 1: the lanes are idle; 2: the first batch since recorded ev_stagger behind its k_minimizer_fast
    bool copies_pending = false;                        // host -> device copies were queued on the context's stream since the last fork
    uint32_t work_lanes = 2;                            // hulk_params.work_lanes
    uint32_t host_copy_threads = 4;                     // hulk_params.host_copy_threads
    bool no_overlap = false, shard_full = false;        // HULK_FLAG_NO_OVERLAP, HULK_FLAG_SHARD_FULL
    bool lds_order_ok = true, cms_chain = false;        // lds_order_verified(device); the count-min replay runs the chain-form kernels (HULK_FLAG_CMS_CHAIN or !lds_order_ok)
    uint64_t *d_long_xs = nullptr, *d_long_table = nullptr; uint8_t *d_long_valid = nullptr;   // long-sequence scratch
    void *d_long_desc = nullptr; uint64_t long_desc_cap = 0;
    // the descriptors of a group of long sequences are built on the host: two pinned staging buffers in turn, each reused when the
    // copy out of it (two groups ago) has run — the calling thread stays up to two groups ahead of the binning (bin_long_reads)
    struct LongDescStage { void *p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool used = false; } h_long_desc[2];
    int long_desc_turn = 0;
    // the descriptors and the set tables on the device are ONE set per context: a group queued on one work lane waits for the group
    // queued before it on the other (the kernels of a group fill the chip: nothing is lost)
    hipEvent_t ev_long = nullptr; hipStream_t long_last_stream = nullptr; bool long_pending = false;
    // A caller that holds the batch's offsets in host memory says so (ctx_hint_host_offsets): the long-sequence path then reads the
    // lengths there instead of fetching them from the device behind everything queued on the lane.  Valid for the next
    // hulk_add_reads_device only; h_off_chunk is that call's current piece.
    const uint64_t *h_off_hint = nullptr, *h_off_chunk = nullptr;
    uint64_t long_cap = 0, long_table_cap = 0;   // minimizer list of the short-read kernel (grow-only)
    // host-side run state
    uint64_t seq_count = 0, flush_index = 0;
    uint32_t T = 16, ring_n = 17, ring_base = 0;   // interval batch size and spectrum ring
    uint32_t bin_spectra = 0;                      // spectra hulk_bin_reads_device filled that no flush has taken yet
    bool tables_ready = false, finished = false, hist_hook_used = false;
    int sticky = HULK_OK;
    std::string last_error;
    int profiling = 0;   /* bit 0 k_cws_scan, bit 1 k_minimizer_fast, bit 2 k_jump_bin + k_jump_left, bit 3 k_cmsd_freq (hulk_set_profiling) */
    std::vector<hulk::ProfileRec> prof;
    std::vector<hulk::ProfMark> marks;
};

namespace hulk {

// ---- hulk_api.hip
const char *err_text(int status);
int fail(hulk_ctx *c, int status, const std::string &extra = std::string());   // sets last_error (NULL ctx: hulk_create's), returns status
int fail_hip(hulk_ctx *c, hipError_t e, const char *what);
#define HIPCHK(c, call)                                             \
    do { hipError_t e_ = (call); if (e_ != hipSuccess) return hulk::fail_hip((c), e_, #call); } while (0)
template <typename T> hipError_t dalloc(T **p, size_t n) { return hipMalloc((void **)p, n ? n * sizeof(T) : sizeof(T)); }

// ---- hulk_tables.hip
int build_chains(hulk_ctx *c);
int install_tables(hulk_ctx *c, const double *r, const double *cc, const double *b);
int generate_tables(hulk_ctx *c);
int ensure_tables(hulk_ctx *c);

// ---- hulk_flush.hip
// The profile events only measure time (hulk_get_profile synchronises the streams before it reads them): without the
// system-scope fence a default event carries, a bracket no longer writes back and invalidates the caches around the kernel
// (k_minimizer_fast's bracket cost 2 % of a C2 step that way, mostly in the kernel behind it)
constexpr unsigned PROFILE_EVENT_FLAGS = hipEventDisableSystemFence;
uint32_t *ring_hist(hulk_ctx *c);
int issue_flush(hulk_ctx *c, hipEvent_t gate = nullptr);
int ring_issue_own_flush(hulk_ctx *c);
hipEvent_t ring_write_event(hulk_ctx *c);
int ring_ready_for_writes(hulk_ctx *c);
int sync_all(hulk_ctx *c);
int fatal_status(hulk_ctx *c);
int bin_reads(hulk_ctx *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n,
              uint32_t max_len, uint64_t bases_bytes, uint64_t interval, uint64_t fill, bool join = false);
hipStream_t lane_stream(hulk_ctx *c, int ring);
hipStream_t ring_stream(hulk_ctx *c);
int lanes_join(hulk_ctx *c);
int lanes_prereserve(hulk_ctx *c);
int stage_mark_busy(hulk_ctx *c, hulk_ctx::HostStage &hs);
int flush_kernels(hulk_ctx *c, hipStream_t s, uint32_t *hist, const FlushBatch &fb);
bool no_overlap_mode(const hulk_ctx *c);
hipStream_t flush_stream_of(hulk_ctx *c);
int flush_batch(hulk_ctx *c, uint32_t count, hipStream_t dep_stream = nullptr, bool use_dep = false, bool allreduce = false);
int check_device_error(hulk_ctx *c);
int stage_host_reads(hulk_ctx *c, const uint8_t *bases, const uint64_t *offsets, uint64_t i0, uint64_t i1,
                     hulk_ctx::HostStage **out);
int check_host_reads(hulk_ctx *c, const uint64_t *offsets, uint64_t n, uint64_t *max_len_out);

// work-lane stream; in the profiling build HULK_K1_CU_FREE=N leaves N of the 256 CUs (every 256/N-th) to the other streams
hipError_t create_lane_stream(hipStream_t *s, int priority);
// arms prof_mark() for the launches this thread issues on behalf of `c` (a context is single-caller)
struct ProfScope {
    hulk_ctx *prev;
    explicit ProfScope(hulk_ctx *c);
    ~ProfScope();
};

// ---- hulk_comm.hip
void comm_teardown(hulk_ctx *c);
int comm_allreduce_u32(hulk_ctx *c, hipStream_t s, uint32_t *d_buf, size_t words);

}  // namespace hulk
