// fake_rccl.cpp -> libfakerccl.so — a TEST DOUBLE for librccl.so.1 (test infrastructure, never shipped, never linked by the
// product).  It exports the eight nccl* entry points hulk_comm.hip binds at run time (hulk_amd/csrc/hulk_comm.hip:17-24) and
// is selected with HULK_RCCL_LIB=<path to this .so>.  Why: RCCL refuses a communicator whose ranks share one GPU
// (profiles/r05_rccl_world2.txt), no multi-GPU node was ever available, and so the library's RCCL branch (transport kind 1:
// grouped in-place all-gathers of payload + header, the in-place uint32 all-reduce, the EOF all-gather of the slot shards)
// had never run with a peer.  With this double, 2..8 PROCESSES on the one GPU drive exactly that branch.
//
// What it keeps of the real thing — the properties the product code relies on or could get wrong:
//   * ASYNCHRONOUS, STREAM-ORDERED: a collective returns at once; it runs on the caller's stream behind the work queued before
//     it and before the work queued after it (D2H copy -> host function -> H2D copy on that stream).  FAKE_RCCL_SYNC=1 makes the
//     call block instead (hipStreamSynchronize around the exchange) — a fallback, not the default.
//   * one communicator = one ordered sequence of operations: an operation issued on ANOTHER stream than the one before it waits
//     for that one (event), as NCCL serialises the operations of a communicator.
//   * ncclGroupStart / ncclGroupEnd: calls inside a group are queued and issued, in order, by the outermost ncclGroupEnd.
//   * in place: ncclAllGather accepts sendbuff == recvbuff + rank * count and ncclAllReduce sendbuff == recvbuff; a send buffer
//     that overlaps the receive buffer anywhere else is ncclInvalidArgument.
//   * ncclCommInitRank is collective (returns when every rank has joined).
// What it is STRICTER about than the real thing: every operation carries {sequence number, kind, bytes}; ranks that disagree
// (one rank took the delta exchange, its peer the spectra exchange; a rank issued one collective more) make every rank print
// both descriptors and abort() — real RCCL would hang or exchange garbage.  A peer that does not arrive within
// FAKE_RCCL_TIMEOUT_S (default 120) is reported the same way.
//
// Transport: POSIX shared memory (one segment per communicator, named in the unique id), one slot of FAKE_RCCL_SLOT_MB
// (default 48) per rank.  Only what hulk uses is implemented: ncclUint8 all-gather, ncclUint32 + ncclSum all-reduce.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace {
constexpr uint32_t MAX_RANKS = 64;
constexpr uint32_t KIND_ALLGATHER = 1, KIND_ALLREDUCE_U32 = 2;

struct RankCell {                                   // one cache line per rank
    std::atomic<uint64_t> posted;                   // sequence number of the latest operation whose data is in this rank's slot
    std::atomic<uint64_t> taken;                    // ... of the latest operation whose slots this rank has finished reading
    uint64_t kind, bytes;                           // descriptor of the operation `posted` names (written before posted)
    uint64_t pad[4];
};
struct Shm {
    std::atomic<uint32_t> joined;
    std::atomic<uint32_t> failed;
    uint32_t pad[14];
    RankCell cell[MAX_RANKS];
};
static_assert(sizeof(RankCell) == 64, "cell");

struct Op {
    struct ncclComm *comm;
    uint32_t kind;
    uint64_t bytes;                                 // per rank
    uint64_t seq;
    const void *send;
    void *recv;
    hipStream_t stream;
};
}  // namespace

struct ncclComm {
    int rank = 0, nranks = 0;
    Shm *shm = nullptr;
    uint8_t *slots = nullptr;                       // nranks x slot_bytes
    size_t slot_bytes = 0, map_bytes = 0;
    uint8_t *stage = nullptr;                       // pinned: [send (bytes)][recv (nranks x bytes)]
    size_t stage_cap = 0;
    uint64_t next_seq = 1;
    hipStream_t last_stream = nullptr;
    bool have_last = false;
    hipEvent_t last_done = nullptr;
    bool sync_mode = false;
    double timeout_s = 120.0;
};

namespace {
thread_local int g_group_depth = 0;
thread_local std::vector<Op> g_group_ops;
std::atomic<uint64_t> g_ops{0}, g_bytes{0}, g_comms{0};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

[[noreturn]] void die(ncclComm *c, const Op &op, const char *what) {
    fprintf(stderr, "fake-rccl: rank %d/%d: %s at operation %llu (kind %u, %llu bytes)\n", c->rank, c->nranks, what,
            (unsigned long long)op.seq, op.kind, (unsigned long long)op.bytes);
    for (int r = 0; r < c->nranks; r++)
        fprintf(stderr, "fake-rccl:   rank %d: posted %llu (kind %llu, %llu bytes), taken %llu\n", r,
                (unsigned long long)c->shm->cell[r].posted.load(), (unsigned long long)c->shm->cell[r].kind,
                (unsigned long long)c->shm->cell[r].bytes, (unsigned long long)c->shm->cell[r].taken.load());
    c->shm->failed.store(1);
    fflush(stderr);
    abort();
}

template <class Pred> bool wait_for(ncclComm *c, Pred p) {
    const double t_end = now_s() + c->timeout_s;
    for (uint32_t spin = 0;; spin++) {
        if (p()) return true;
        if (c->shm->failed.load(std::memory_order_relaxed)) return false;
        if ((spin & 63) == 63) { if (now_s() > t_end) return false; usleep(50); }
        else sched_yield();
    }
}

// the exchange itself, on the host: stage[0, bytes) -> every rank's stage[bytes, bytes + result)
void exchange(const Op &op) {
    ncclComm *c = op.comm;
    Shm *s = c->shm;
    const int R = c->nranks, me = c->rank;
    // 1. nobody still reads the slots of the operation before this one
    if (!wait_for(c, [&] { for (int r = 0; r < R; r++) if (s->cell[r].taken.load(std::memory_order_acquire) + 1 < op.seq) return false; return true; }))
        die(c, op, "a peer has not finished the PREVIOUS operation (timeout or failed peer)");
    // 2. post: descriptor, data, then the sequence number
    s->cell[me].kind = op.kind; s->cell[me].bytes = op.bytes;
    memcpy(c->slots + (size_t)me * c->slot_bytes, c->stage, op.bytes);
    s->cell[me].posted.store(op.seq, std::memory_order_release);
    // 3. everyone has posted THIS operation — and it is the same operation everywhere
    if (!wait_for(c, [&] { for (int r = 0; r < R; r++) if (s->cell[r].posted.load(std::memory_order_acquire) < op.seq) return false; return true; }))
        die(c, op, "a peer has not issued this operation (timeout or failed peer)");
    for (int r = 0; r < R; r++) {
        if (s->cell[r].posted.load(std::memory_order_acquire) != op.seq) die(c, op, "a peer is AHEAD of this rank (it issued more operations)");
        if (s->cell[r].kind != op.kind || s->cell[r].bytes != op.bytes) die(c, op, "the ranks disagree on the operation (kind / size): they are out of step");
    }
    // 4. read
    uint8_t *out = c->stage + op.bytes;
    if (op.kind == KIND_ALLGATHER) {
        for (int r = 0; r < R; r++) memcpy(out + (size_t)r * op.bytes, c->slots + (size_t)r * c->slot_bytes, op.bytes);
    } else {
        const size_t n = op.bytes / 4;
        uint32_t *acc = (uint32_t *)out;
        memcpy(acc, c->slots, op.bytes);
        for (int r = 1; r < R; r++) {
            const uint32_t *p = (const uint32_t *)(c->slots + (size_t)r * c->slot_bytes);
            for (size_t i = 0; i < n; i++) acc[i] += p[i];
        }
    }
    s->cell[me].taken.store(op.seq, std::memory_order_release);
    g_ops.fetch_add(1); g_bytes.fetch_add(op.bytes);
}

void host_fn(void *p) {
    Op *op = (Op *)p;
    exchange(*op);
    delete op;
}

ncclResult_t issue(const Op &op0) {
    ncclComm *c = op0.comm;
    Op op = op0;
    op.seq = c->next_seq++;
    const size_t out_bytes = op.kind == KIND_ALLGATHER ? (size_t)c->nranks * op.bytes : op.bytes;
    const size_t need = op.bytes + out_bytes;
    if (op.bytes > c->slot_bytes) {
        fprintf(stderr, "fake-rccl: %llu bytes per rank exceed the slot (FAKE_RCCL_SLOT_MB)\n", (unsigned long long)op.bytes);
        return ncclInvalidArgument;
    }
    // a communicator's operations are one sequence: an operation on another stream waits for the one before it
    if (c->have_last && c->last_stream != op.stream && hipStreamWaitEvent(op.stream, c->last_done, 0) != hipSuccess) return ncclUnhandledCudaError;
    if (need > c->stage_cap) {                      // (growing the staging: nothing of this communicator may be in flight)
        if (c->have_last && hipEventSynchronize(c->last_done) != hipSuccess) return ncclUnhandledCudaError;
        if (c->stage) (void)hipHostFree(c->stage);
        c->stage = nullptr; c->stage_cap = 0;
        if (hipHostMalloc((void **)&c->stage, need + need / 2, hipHostMallocDefault) != hipSuccess) return ncclUnhandledCudaError;
        c->stage_cap = need + need / 2;
    }
    if (hipMemcpyAsync(c->stage, op.send, op.bytes, hipMemcpyDeviceToHost, op.stream) != hipSuccess) return ncclUnhandledCudaError;
    if (c->sync_mode) {
        if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
        exchange(op);
    } else {
        Op *heap = new Op(op);
        if (hipLaunchHostFunc(op.stream, host_fn, heap) != hipSuccess) { delete heap; return ncclUnhandledCudaError; }
    }
    if (hipMemcpyAsync(op.recv, c->stage + op.bytes, out_bytes, hipMemcpyHostToDevice, op.stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventRecord(c->last_done, op.stream) != hipSuccess) return ncclUnhandledCudaError;
    c->last_stream = op.stream; c->have_last = true;
    return ncclSuccess;
}

ncclResult_t submit(const Op &op) {
    if (g_group_depth > 0) { g_group_ops.push_back(op); return ncclSuccess; }
    return issue(op);
}

bool overlaps(const void *a, size_t na, const void *b, size_t nb) {
    const uintptr_t x = (uintptr_t)a, y = (uintptr_t)b;
    return x < y + nb && y < x + na;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    static std::atomic<uint32_t> n{0};
    snprintf(id->internal, sizeof id->internal, "/fakerccl-%d-%u-%llx", (int)getpid(), n.fetch_add(1),
             (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > (int)MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if (strncmp(id.internal, "/fakerccl-", 10) != 0) return ncclInvalidArgument;
    id.internal[sizeof id.internal - 1] = 0;
    ncclComm *c = new ncclComm;
    c->rank = rank; c->nranks = nranks;
    const char *e;
    c->slot_bytes = (size_t)((e = getenv("FAKE_RCCL_SLOT_MB")) ? atol(e) : 48) << 20;
    c->timeout_s = (e = getenv("FAKE_RCCL_TIMEOUT_S")) ? atof(e) : 120.0;
    c->sync_mode = (e = getenv("FAKE_RCCL_SYNC")) && atoi(e) != 0;
    const size_t hdr_bytes = (sizeof(Shm) + 4095) & ~(size_t)4095;
    c->map_bytes = hdr_bytes + (size_t)nranks * c->slot_bytes;
    const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { perror("fake-rccl: shm_open"); delete c; return ncclSystemError; }
    if (ftruncate(fd, (off_t)c->map_bytes) != 0) { perror("fake-rccl: ftruncate"); close(fd); delete c; return ncclSystemError; }
    void *m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { perror("fake-rccl: mmap"); delete c; return ncclSystemError; }
    c->shm = (Shm *)m;                                                    // (a fresh segment is zero: every atomic starts at 0)
    c->slots = (uint8_t *)m + hdr_bytes;
    if (hipEventCreateWithFlags(&c->last_done, hipEventDisableTiming) != hipSuccess) { munmap(m, c->map_bytes); delete c; return ncclUnhandledCudaError; }
    c->shm->joined.fetch_add(1);
    const bool all = wait_for(c, [&] { return c->shm->joined.load() >= (uint32_t)nranks; });
    if (rank == 0 || !all) shm_unlink(id.internal);                       // everyone holds a mapping (or the join failed): the name can go
    if (!all) {
        fprintf(stderr, "fake-rccl: rank %d/%d: only %u ranks joined the communicator within %.0f s\n", rank, nranks, c->shm->joined.load(), c->timeout_s);
        (void)hipEventDestroy(c->last_done); munmap(m, c->map_bytes); delete c;
        return ncclSystemError;
    }
    if (getenv("FAKE_RCCL_VERBOSE")) fprintf(stderr, "fake-rccl: rank %d/%d joined %s (%s)\n", rank, nranks, id.internal, c->sync_mode ? "blocking" : "asynchronous");
    g_comms.fetch_add(1);
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclInvalidArgument;
    if (c->have_last) (void)hipEventSynchronize(c->last_done);
    if (c->last_done) (void)hipEventDestroy(c->last_done);
    if (c->stage) (void)hipHostFree(c->stage);
    if (c->shm) munmap((void *)c->shm, c->map_bytes);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
    if (!comm || !sendbuff || !recvbuff) return ncclInvalidArgument;
    if (datatype != ncclUint8 && datatype != ncclInt8) { fprintf(stderr, "fake-rccl: ncclAllGather: only 1-byte types\n"); return ncclInvalidArgument; }
    const uint8_t *own = (const uint8_t *)recvbuff + (size_t)comm->rank * sendcount;
    if ((const uint8_t *)sendbuff != own && overlaps(sendbuff, sendcount, recvbuff, sendcount * comm->nranks)) {
        fprintf(stderr, "fake-rccl: ncclAllGather: sendbuff overlaps recvbuff but is not recvbuff + rank * sendcount\n");
        return ncclInvalidArgument;
    }
    if (sendcount == 0) return ncclSuccess;
    Op op{comm, KIND_ALLGATHER, sendcount, 0, sendbuff, recvbuff, stream};
    return submit(op);
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op_, ncclComm_t comm, hipStream_t stream) {
    if (!comm || !sendbuff || !recvbuff) return ncclInvalidArgument;
    if (datatype != ncclUint32 || op_ != ncclSum) { fprintf(stderr, "fake-rccl: ncclAllReduce: only ncclUint32 + ncclSum\n"); return ncclInvalidArgument; }
    if (sendbuff != recvbuff && overlaps(sendbuff, count * 4, recvbuff, count * 4)) return ncclInvalidArgument;
    if (count == 0) return ncclSuccess;
    Op op{comm, KIND_ALLREDUCE_U32, count * 4, 0, sendbuff, recvbuff, stream};
    return submit(op);
}

ncclResult_t ncclGroupStart() { g_group_depth++; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
    if (g_group_depth <= 0) return ncclInvalidUsage;
    if (--g_group_depth > 0) return ncclSuccess;
    ncclResult_t rc = ncclSuccess;
    for (const Op &op : g_group_ops) { const ncclResult_t r = issue(op); if (r != ncclSuccess && rc == ncclSuccess) rc = r; }
    g_group_ops.clear();
    return rc;
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "fake-rccl: no error";
        case ncclUnhandledCudaError: return "fake-rccl: unhandled HIP error";
        case ncclSystemError: return "fake-rccl: system error";
        case ncclInvalidArgument: return "fake-rccl: invalid argument";
        case ncclInvalidUsage: return "fake-rccl: invalid usage";
        default: return "fake-rccl: error";
    }
}

// not part of RCCL: lets a test assert that THIS library carried the traffic (communicators built, collectives run, bytes per rank sent)
void fakeRcclStats(uint64_t *comms, uint64_t *ops, uint64_t *bytes) {
    if (comms) *comms = g_comms.load();
    if (ops) *ops = g_ops.load();
    if (bytes) *bytes = g_bytes.load();
}

}  // extern "C"
