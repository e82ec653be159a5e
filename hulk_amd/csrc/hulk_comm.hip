// hulk_comm.hip — multi-GPU with the exchange INSIDE the library (include/hulk_hip.h): RCCL bound at run time, the host
// and loopback transports, hulk_step_sharded / hulk_step_sliced, hulk_gather_sketch.  Reference seam: SeqMinimizer.Run's
// AddSeq / Flush loop (src/pipeline/sketch.go:182-250) with the read stream sharded over one process per GPU.
#include "hulk_ctx.h"

#include <dlfcn.h>

#include <algorithm>

namespace hulk {
namespace {
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
// hulk_debug_inject (test hook of the profiling build; the shipping library has neither the entry point nor the branches)
#ifdef HULK_EXPERIMENTS
inline bool injected(const hulk_ctx::Comm &m, uint32_t what) { return m.inject == what && m.inject_step == m.step; }
#else
inline bool injected(const hulk_ctx::Comm &, uint32_t) { return false; }
constexpr uint32_t HULK_INJECT_STALE_SEAL = 1u, HULK_INJECT_STALE_STAGE = 2u;
#endif
int fail_nccl(hulk_ctx *c, ncclResult_t r, const char *what) {
    return fail(c, HULK_ERR_COMM, std::string(what) + ": " + (rccl()->GetErrorString ? rccl()->GetErrorString(r) : "RCCL error"));
}
#define NCCLCHK(c, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return fail_nccl((c), r_, #call); } while (0)

}  // namespace

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
    if (c->flush_stream) hipStreamSynchronize(c->flush_stream);   // no collective in flight when the communicator goes
    if (c->stream) hipStreamSynchronize(c->stream);
    if (m.nccl && rccl()->CommDestroy) rccl()->CommDestroy(m.nccl);
    hipFree(m.d_hdr); hipFree(m.d_delta); hipFree(m.d_gather); hipFree(m.d_sk);
    for (int i = 0; i < 2; i++) { if (m.h_hdr[i]) hipHostFree(m.h_hdr[i]); if (m.ev_hdr[i]) hipEventDestroy(m.ev_hdr[i]); }
    if (m.h_stage) hipHostFree(m.h_stage);
    m = hulk_ctx::Comm{};
}

// The exchange of a sharded step on stream s: every rank's payload (count-min increments or spectra; may be empty) and,
// BEHIND it, its sealed header block.  The collectives run on the flush stream itself, in line with the kernels that wrote
// the buffers and the kernels that read them: the dependency is the stream's order and nothing else.  (Round 4 ran them on
// a highest-priority stream of their own behind an event, and met a void header block once per ~20 runs of the in-process
// fuzz under GPU_MAX_HW_QUEUES=16.  The bare pattern — fills + kernel -> event -> copy on another stream — does not
// misorder on this runtime: 3.9e6 iterations of tools/ubench/hdr_race.hip, profiles/r05_hdr_race.txt.  One stream fewer
// per rank, no cross-stream edge to get wrong, and a header that proves its own integrity: see the seal.)
int comm_exchange(hulk_ctx *c, hipStream_t s, uint32_t *own_hdr, const void *own_payload, void *d_payload_all, size_t pbytes,
                  uint32_t step_tag) {
    hulk_ctx::Comm &m = c->comm;
    const size_t hbytes = SHARD_HDR * 4;
    m.bytes_rx += (uint64_t)(pbytes + hbytes) * (m.world - 1);
    switch (m.kind) {
        case 1: {
            NCCLCHK(c, rccl()->GroupStart());
            ncclResult_t r1 = ncclSuccess, r2 = ncclSuccess;
            if (pbytes) r1 = rccl()->AllGather(own_payload, d_payload_all, pbytes, ncclUint8, m.nccl, s);
            r2 = rccl()->AllGather(own_hdr, m.d_hdr, hbytes, ncclUint8, m.nccl, s);
            const ncclResult_t r3 = rccl()->GroupEnd();              // (closed whatever the calls inside it returned)
            if (r1 != ncclSuccess) return fail_nccl(c, r1, "ncclAllGather (payload)");
            if (r2 != ncclSuccess) return fail_nccl(c, r2, "ncclAllGather (header)");
            if (r3 != ncclSuccess) return fail_nccl(c, r3, "ncclGroupEnd");
            return HULK_OK;
        }
        case 2: {
            // staging: [own payload][own header][gathered payloads][gathered headers].  The header is copied out BEHIND the
            // payload on the same stream and checked on the host before anything is sent: this rank knows the seal its block
            // must carry.  A copy that is not there after the stream's synchronisation (seen once per ~20 runs of the 4..8-rank
            // in-process fuzz with 16 hardware queues) is waited for and taken again.
            { const int rc = comm_host_stage(c, (pbytes + hbytes) * (m.world + 1)); if (rc != HULK_OK) return rc; }
            uint8_t *h_pay = m.h_stage, *h_hdr = m.h_stage + pbytes, *h_pay_all = h_hdr + hbytes, *h_hdr_all = h_pay_all + pbytes * m.world;
            const volatile uint32_t *hv = (const volatile uint32_t *)h_hdr;
            for (int attempt = 0;; attempt++) {
                ((volatile uint32_t *)h_hdr)[SHARD_TAG] = 0;
                const bool skip = injected(m, HULK_INJECT_STALE_STAGE) && attempt == 0;   // test hook
                if (pbytes) HIPCHK(c, hipMemcpyAsync(h_pay, own_payload, pbytes, hipMemcpyDeviceToHost, s));
                if (!skip) HIPCHK(c, hipMemcpyAsync(h_hdr, own_hdr, hbytes, hipMemcpyDeviceToHost, s));
                HIPCHK(c, hipStreamSynchronize(s));
                if (hv[SHARD_TAG] == step_tag || injected(m, HULK_INJECT_STALE_SEAL)) break;      // (the hook's void seal is this step's only)
                m.hdr_resyncs++;
                if (attempt == 3) return fail(c, HULK_ERR_COMM, "this rank's exchange header did not reach its host staging (4 attempts)");
                HIPCHK(c, hipDeviceSynchronize());
            }
            if (pbytes && m.fn(m.user, HULK_XCHG_ALLGATHER, h_pay, h_pay_all, pbytes) != 0)
                return fail(c, HULK_ERR_COMM, "the host's exchange function failed (all-gather)");
            if (m.fn(m.user, HULK_XCHG_ALLGATHER, h_hdr, h_hdr_all, hbytes) != 0)
                return fail(c, HULK_ERR_COMM, "the host's exchange function failed (all-gather)");
            if (pbytes) HIPCHK(c, hipMemcpyAsync(d_payload_all, h_pay_all, pbytes * m.world, hipMemcpyHostToDevice, s));
            HIPCHK(c, hipMemcpyAsync(m.d_hdr, h_hdr_all, hbytes * m.world, hipMemcpyHostToDevice, s));
            HIPCHK(c, hipStreamSynchronize(s));
            return HULK_OK;
        }
        case 3:
            for (uint32_t r = 0; r < m.world; r++) {
                uint8_t *dp = (uint8_t *)d_payload_all + (size_t)r * pbytes;
                uint32_t *dh = m.d_hdr + (size_t)r * SHARD_HDR;
                if (pbytes && dp != (const uint8_t *)own_payload) HIPCHK(c, hipMemcpyAsync(dp, own_payload, pbytes, hipMemcpyDeviceToDevice, s));
                if (dh != own_hdr) HIPCHK(c, hipMemcpyAsync(dh, own_hdr, hbytes, hipMemcpyDeviceToDevice, s));
            }
            return HULK_OK;
        default: break;
    }
    return fail(c, HULK_ERR_STATE, "no communicator");
}

namespace {
int comm_setup_alloc(hulk_ctx *c, int kind, uint32_t rank, uint32_t world);
// (a failure half-way leaves nothing behind: the call can be repeated — ADVICE r3)
int comm_setup(hulk_ctx *c, int kind, uint32_t rank, uint32_t world) {
    hulk_ctx::Comm &m = c->comm;
    if (m.kind != 0) return fail(c, HULK_ERR_STATE, "the context already has a communicator");
    if (world == 0 || rank >= world) return fail(c, HULK_ERR_ARG, "rank / world");
    if (c->seq_count || c->flush_index) return fail(c, HULK_ERR_STATE, "hulk_comm_init must precede the first read");
    const int rc = comm_setup_alloc(c, kind, rank, world);
    if (rc != HULK_OK) { const std::string keep = c->last_error; comm_teardown(c); c->last_error = keep; }
    return rc;
}
int comm_setup_alloc(hulk_ctx *c, int kind, uint32_t rank, uint32_t world) {
    hulk_ctx::Comm &m = c->comm;
    HIPCHK(c, hipSetDevice(c->p.device));
    const size_t NC = (size_t)c->cms_depth * c->cms_width;
    HIPCHK(c, dalloc(&m.d_hdr, (size_t)world * SHARD_HDR));
    HIPCHK(c, dalloc(&m.d_delta, (size_t)world * c->T * NC));
    HIPCHK(c, dalloc(&m.d_sk, (size_t)world * (2 + 2 * (size_t)c->S)));
    HIPCHK(c, hipMemset(m.d_hdr, 0, (size_t)world * SHARD_HDR * 4));
    HIPCHK(c, hipDeviceSynchronize());
    for (int i = 0; i < 2; i++) {
        // the host's view of a step's gathered header: mapped pinned memory, stored to by k_shard_check; its event is a
        // default one (system-scope release when it completes)
        HIPCHK(c, hipHostMalloc((void **)&m.h_hdr[i], (size_t)world * SHARD_HDR * 4, hipHostMallocMapped));
        memset(m.h_hdr[i], 0, (size_t)world * SHARD_HDR * 4);
        HIPCHK(c, hipEventCreate(&m.ev_hdr[i]));
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
}  // namespace hulk

using namespace hulk;

extern "C" {
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

static int step_sharded_impl(hulk_ctx *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n, uint32_t max_read_len,
                             uint64_t bases_bytes, uint32_t step_intervals);
int hulk_step_sharded(hulk_ctx *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n, uint32_t max_read_len,
                      uint64_t bases_bytes, uint32_t step_intervals) {
    if (!c) return HULK_ERR_ARG;
    const int rc = step_sharded_impl(c, d_bases, d_offsets, n, max_read_len, bases_bytes, step_intervals);
    // a runtime or exchange failure may have left this rank outside a collective its peers are inside of: the communicator is
    // not usable any more, and neither is the run — every later call reports the same status (argument errors, raised before
    // anything is queued, leave the context as it was)
    if (rc == HULK_ERR_COMM || rc == HULK_ERR_HIP) c->sticky = rc;
    return rc;
}
static int step_sharded_impl(hulk_ctx *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n, uint32_t max_read_len,
                             uint64_t bases_bytes, uint32_t step_intervals) {
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
    hipStream_t s = flush_stream_of(c);
    const uint32_t tag = (uint32_t)(m.step + 1);                    // the seal of this step's header blocks
    // 2. which exchange: the verdicts of the step before (they travelled with its exchange) — any rank's need_full keeps
    //    the spectra exchange.  The wait ends when the previous step's exchange has run: this step's binning is queued.
    bool full = c->drift || c->scaling || !c->prune || c->no_skip || m.step == 0;
    if (!full) {
        const int prev = (int)((m.step - 1) & 1);
        if (m.hdr_pending[prev]) { HIPCHK(c, hipEventSynchronize(m.ev_hdr[prev])); m.hdr_pending[prev] = false; }
        // every block of the gathered header is sealed with the number of the step it was written in: what is read here
        // must be the previous step's, from every rank — a rank that read an older copy of the buffer would pick another
        // exchange than its peers, which no collective survives
        const volatile uint32_t *hh = m.h_hdr[prev];
        auto stale = [&] { for (uint32_t r = 0; r < m.world; r++) if (hh[(size_t)r * SHARD_HDR + SHARD_TAG] != (uint32_t)m.step) return true; return false; };
        if (stale()) {
            // the device buffer still holds the previous step's gathered header (this step's writes are not queued yet):
            // wait for the stream and fetch it with a copy
            HIPCHK(c, hipStreamSynchronize(s));
            HIPCHK(c, hipMemcpyAsync(m.h_hdr[prev], m.d_hdr, (size_t)m.world * SHARD_HDR * 4, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipStreamSynchronize(s));
            m.hdr_resyncs++;
            // still not the previous step's: a block was void when its rank SENT it — every rank holds the same gathered
            // bytes and comes to the same conclusion: the verdicts are unknown, and the spectra exchange is always right.
            // (If the previous step took the delta exchange, k_shard_check has raised HULK_ERR_COMM on every rank as well.)
            if (stale()) { full = true; m.hdr_void++; }
        }
        for (uint32_t r = 0; r < m.world; r++) if (m.h_hdr[prev][(size_t)r * SHARD_HDR + SHARD_VERDICT]) full = true;
    }
    if (c->shard_full) full = true;                                           // HULK_FLAG_SHARD_FULL: always the spectra exchange
    HIPCHK(c, hipEventRecord(c->ev_binned, ring_stream(c)));        // (the binning ran on the ring's work lane)
    if (!no_overlap_mode(c)) HIPCHK(c, hipStreamWaitEvent(s, c->ev_binned, 0));
    const int ring = c->cur_ring;
    uint32_t *hist = ring_hist(c);
    const size_t B = (size_t)c->B, NC = (size_t)c->cms_depth * c->cms_width;
    uint32_t *own_hdr = m.d_hdr + (size_t)m.rank * SHARD_HDR;
    FlushBatch fb{};
    fb.ring_base = 0; fb.ring_n = c->ring_n; fb.count = own; fb.parity = 0; fb.num_bins = c->B;
    HIPCHK(c, hipMemsetAsync(own_hdr, 0, SHARD_HDR * 4, s));
    uint32_t *own_delta = m.d_delta + (size_t)m.rank * c->T * NC;
    if (!full) {
        HIPCHK(c, hipMemsetAsync(own_delta, 0, (size_t)c->T * NC * 4, s));
        HIPCHK(c, launch_shard_local(s, hist, c->d_pos16, own_hdr, own_delta, c->cms_depth, c->cms_width, fb));
        HIPCHK(c, hipEventRecord(c->ev_flushed[ring], s));          // the ring is wiped: the work stream may fill it again
        c->pending_flush[ring] = true;
    }
    // this rank's verdict for the NEXT step — the whole-batch bound on the counters and weights as they stand now (a rank
    // without slots has nothing to protect: verdict 0) — sealed into the block with the step's tag by the LAST store into it
    const bool void_seal = injected(m, HULK_INJECT_STALE_SEAL);    // test hook: a block of another step
    HIPCHK(c, launch_flush_decide(s, c->d_ctr, (int)NC, c->d_kminslot, c->d_weights, (int)c->slots, (int)c->slot_begin,
                                  c->d_state, fb, c->slots ? 1 : 2, (unsigned long long *)own_hdr, void_seal ? tag - 1 : tag));
    const int cur = (int)(m.step & 1);
    if (!full) {
        rc = comm_exchange(c, s, own_hdr, own_delta, m.d_delta, (size_t)c->T * NC * 4, tag);
        if (rc != HULK_OK) return rc;
        // a void block in a delta step cannot be repaired (the spectra are wiped): HULK_ERR_COMM on every rank
        HIPCHK(c, launch_shard_check(s, m.d_hdr, m.world, tag, c->d_state, m.h_hdr[cur], 1));
        HIPCHK(c, hipEventRecord(m.ev_hdr[cur], s));
        HIPCHK(c, launch_shard_apply(s, m.d_hdr, m.d_delta, c->d_ctr, c->cms_depth, c->cms_width, m.world, c->T,
                                     step_intervals, c->B, c->d_state, tag));
        m.steps_delta++;
    } else {
        const size_t need = (size_t)m.world * c->T * B;
        if (need > m.gather_words) {
            HIPCHK(c, hipStreamSynchronize(s));
            hipFree(m.d_gather); m.d_gather = nullptr; m.gather_words = 0;
            HIPCHK(c, hipMalloc((void **)&m.d_gather, need * 4));
            m.gather_words = need;
        }
        rc = comm_exchange(c, s, own_hdr, hist, m.d_gather, (size_t)c->T * B * 4, tag);
        if (rc != HULK_OK) return rc;
        // (the header of a spectra exchange carries only the verdicts: a void block makes the next step a spectra exchange)
        HIPCHK(c, launch_shard_check(s, m.d_hdr, m.world, tag, c->d_state, m.h_hdr[cur], 0));
        HIPCHK(c, hipEventRecord(m.ev_hdr[cur], s));
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
    // (the host's view of the gathered header was stored by k_shard_check and its event recorded right behind that kernel:
    // the flush kernels queued after it do not delay the next step's choice)
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
    { const int rcb = stage_mark_busy(c, *hs); if (rcb != HULK_OK) return rcb; }      // (copies on the context's stream, kernels on a work lane)
    return rc;
}

int hulk_step_sliced(hulk_ctx *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n, uint32_t max_read_len,
                     uint64_t bases_bytes, uint64_t reads_per_spectrum, uint32_t n_spectra) {
    if (!c) return HULK_ERR_ARG;
    if (c->comm.kind == 0) return fail(c, HULK_ERR_STATE, "hulk_step_sliced needs hulk_comm_init");
    if (n_spectra == 0 || n_spectra > c->T) return fail(c, HULK_ERR_ARG, "n_spectra");
    int rc = hulk_bin_reads_device_at(c, d_bases, d_offsets, n, max_read_len, bases_bytes, reads_per_spectrum, 0);
    if (rc == HULK_OK && c->bin_spectra > n_spectra) return fail(c, HULK_ERR_ARG, "more spectra binned than n_spectra");
    if (rc == HULK_OK) rc = flush_batch(c, n_spectra, nullptr, false, true);
    if (rc == HULK_OK) { c->cur_ring ^= 1; c->bin_spectra = 0; }
    if (rc == HULK_ERR_COMM || rc == HULK_ERR_HIP) c->sticky = rc;      // as hulk_step_sharded: the peers may be inside the all-reduce
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

int hulk_get_comm_health(hulk_ctx *c, uint64_t *refetched, uint64_t *void_blocks) {
    if (!c) return HULK_ERR_ARG;
    if (refetched) *refetched = c->comm.hdr_resyncs;
    if (void_blocks) *void_blocks = c->comm.hdr_void;
    return HULK_OK;
}

#ifdef HULK_EXPERIMENTS
int hulk_debug_inject(hulk_ctx *c, uint32_t what, uint64_t step) {
    if (!c) return HULK_ERR_ARG;
    if (what > HULK_INJECT_STALE_STAGE) return fail(c, HULK_ERR_ARG, "hulk_debug_inject: what");
    c->comm.inject = what; c->comm.inject_step = step;
    return HULK_OK;
}
#endif

int hulk_get_comm_stats(hulk_ctx *c, uint64_t *steps_delta, uint64_t *steps_full, uint64_t *bytes_received) {
    if (!c) return HULK_ERR_ARG;
    if (steps_delta) *steps_delta = c->comm.steps_delta;
    if (steps_full) *steps_full = c->comm.steps_full;
    if (bytes_received) *bytes_received = c->comm.bytes_rx;
    return HULK_OK;
}

}  // extern "C"
