// Host ingest of libhulkhip: files / STDIN -> lines -> sequences -> pinned staging -> HBM.
//
// Replaces, with the same observable semantics, the reference's
//   DataStreamer.Run   src/pipeline/sketch.go:40-79   (bufio.Scanner lines; gzip when the name ends
//                                                       in ".gz"; STDIN when no file is given)
//   FastqHandler.Run   src/pipeline/sketch.go:99-161  (four nil-tested line slots; FASTA branch)
//   seqio.NewFASTQread src/seqio/seqio.go:38-40       ('@' check when the 4th line arrives)
// and the AddSeq loop of SeqMinimizer.Run (sketch.go:196-217) when a context is attached.
//
// Shape: a reader thread turns the inputs into 32 MB blocks that end on a line boundary (gzip
// inflation runs there, ahead of the parser); a block is parsed by P threads in two passes — pass 1
// runs the 4-state line machine for all four possible start states of every piece (state,
// sequences, bytes), a serial prefix fixes each piece's real start state and its output offsets,
// pass 2 copies the sequence lines straight into pinned staging — and is handed to the GPU with
// asynchronous copies on the context's stream while the next block is read and parsed.
#include <errno.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hulk_hip.h"
#include "hulk_internal.h"
#include "fast_inflate.h"
#include "par_inflate.h"
#include "crc32_clmul.h"
#include "hulk_fastq.h"

namespace {

constexpr size_t MAX_TOKEN = 64 * 1024;      // bufio.MaxScanTokenSize
// The knobs of ONE run (hulk_ingest_opts, include/hulk_hip.h): resolved once by run_ingest — defaults, then the caller's
// fields, then the HULK_* environment variables as overrides for profiling scripts and tests — and handed down to the
// readers; two runs side by side (two contexts of one host process) do not share them.
struct IngestCfg {
    size_t block = (size_t)(32u << 20);       // bytes per block of the line pump (>= 128 KiB)
    unsigned parser_threads = 0;              // 0: one per hardware thread, at most 16 (more were measured slower)
    unsigned gz_threads = 16;                 // members of a bgzip'd input / chunks of ONE gzip member inflated side by side
    bool gz_par = true;                       // one ordinary gzip member on gz_threads threads (GzPar)
    size_t gz_chunk = (size_t)(1u << 20);     // GzPar: compressed bytes per chunk (>= 8 KiB)
    unsigned readers = 4;                     // pieces a block of a regular file is pread() in, side by side
    bool zlib = false;                        // zlib's inflate instead of fast_inflate.h
    bool trace = false;                       // per-phase seconds on stderr
    bool host_parser = false;                 // FASTQ lines -> reads on the host's parser threads instead of the device
    bool block_set = false, readers_set = false;   // the caller (or the environment) chose; else the device path takes its own defaults
};
static IngestCfg resolve_cfg(const hulk_ingest_opts *o, uint32_t threads) {
    IngestCfg c;
    c.parser_threads = threads;
    if (o) {
        if (o->parser_threads) c.parser_threads = o->parser_threads;
        if (o->gz_threads) c.gz_threads = o->gz_threads;
        if (o->block_bytes) { c.block = (size_t)o->block_bytes; c.block_set = true; }
        if (o->gz_chunk_bytes) c.gz_chunk = (size_t)o->gz_chunk_bytes;
        if (o->file_readers) { c.readers = o->file_readers; c.readers_set = true; }
        if (o->flags & HULK_INGEST_GZ_ONE_THREAD) c.gz_par = false;
        if (o->flags & HULK_INGEST_GZ_ZLIB) c.zlib = true;
        if (o->flags & HULK_INGEST_TRACE) c.trace = true;
        if (o->flags & HULK_INGEST_HOST_PARSER) c.host_parser = true;
    }
    if (!o || !o->gz_threads) {                   // the default of 16 inflate threads is for hosts that have them
        const long hw = (long)std::thread::hardware_concurrency();
        if (hw > 0 && (long)c.gz_threads > hw) c.gz_threads = (unsigned)hw;
    }
    if (const char *e = getenv("HULK_INGEST_BLOCK")) { c.block = (size_t)strtoull(e, nullptr, 10); c.block_set = true; }
    if (const char *e = getenv("HULK_GZ_THREADS")) c.gz_threads = (unsigned)std::max(1L, strtol(e, nullptr, 10));
    if (const char *e = getenv("HULK_GZ_PAR")) c.gz_par = !(e[0] == '0');
    if (const char *e = getenv("HULK_GZ_PAR_CHUNK")) c.gz_chunk = (size_t)strtoull(e, nullptr, 10);
    if (const char *e = getenv("HULK_INGEST_READERS")) { c.readers = (unsigned)std::max(1L, strtol(e, nullptr, 10)); c.readers_set = true; }
    if (getenv("HULK_GZ_ZLIB")) c.zlib = true;
    if (getenv("HULK_INGEST_TRACE")) c.trace = true;
    if (c.block < 2 * MAX_TOKEN) c.block = 2 * MAX_TOKEN;
    if (c.gz_threads < 1) c.gz_threads = 1;
    if (c.gz_threads > 64) c.gz_threads = 64;
    if (c.gz_chunk < (8u << 10)) c.gz_chunk = 8u << 10;
    if (c.readers < 1) c.readers = 1;
    if (c.readers > 16) c.readers = 16;
    return c;
}
// hulk_ingest_opts as hulk_create checks hulk_params: unknown flags, non-zero reserved fields and values outside the ranges
// the header states are refused, not clamped (an empty string: the options are fine)
static std::string check_opts(const hulk_ingest_opts *o) {
    if (!o) return std::string();
    if (o->flags & ~(HULK_INGEST_GZ_ONE_THREAD | HULK_INGEST_GZ_ZLIB | HULK_INGEST_TRACE | HULK_INGEST_HOST_PARSER)) return "hulk_ingest_opts: unknown flags";
    if (o->reserved[0] || o->reserved[1]) return "hulk_ingest_opts: reserved must be 0";
    if (o->parser_threads > 256) return "hulk_ingest_opts: parser_threads must be 0 (default) or 1..256";
    if (o->gz_threads > 64) return "hulk_ingest_opts: gz_threads must be 0 (default) or 1..64";
    if (o->file_readers > 16) return "hulk_ingest_opts: file_readers must be 0 (default) or 1..16";
    if (o->block_bytes && o->block_bytes < 2 * MAX_TOKEN) return "hulk_ingest_opts: block_bytes must be 0 (default) or >= 128 KiB";
    if (o->block_bytes > (1ull << 31)) return "hulk_ingest_opts: block_bytes must be <= 2 GiB";
    if (o->gz_chunk_bytes && o->gz_chunk_bytes < (8u << 10)) return "hulk_ingest_opts: gz_chunk_bytes must be 0 (default) or >= 8 KiB";
    return std::string();
}
constexpr size_t FASTA_BATCH_BYTES = 64u << 20;

struct IngestError {
    int code = HULK_OK;
    std::string msg;
    bool set(int c, const std::string &m) { if (code == HULK_OK) { code = c; msg = m; } return false; }
};

// ------------------------------------------------------------------------------------------
// gzip reader on hulk::inflate (fast_inflate.h).  A producer thread inflates into 1 MB chunks (each
// with the previous chunk's last 32 KiB in front of it as match history); the consumer — the block
// reader's thread — takes the CRC-32 of a chunk (crc32_clmul.h) while copying it out, so that inflate
// and checksum run side by side.  Members are concatenated (compress/gzip's multistream default);
// what follows the last member and is not a gzip header is ignored, as zlib's gzread does.
// HULK_GZ_ZLIB=1 keeps zlib's inflate (gzread) instead.
// ------------------------------------------------------------------------------------------
struct GzStream {
    virtual ~GzStream() {}
    virtual long read(uint8_t *dst, size_t cap, std::string &msg) = 0;      // up to cap bytes; 0 = end of the stream; -1 = error (msg filled)
};

// where a reader that inflated the front of a member by other means (GzPar) hands the member over: the descriptor is positioned
// at the byte that holds the first bit of a block header, `bit` of its bits belong to the block in front; the last <= 32 KiB
// of text, and the CRC-32 / length of all the member's text so far
struct GzResume { int bit = 0; std::vector<uint8_t> hist; uint32_t crc = 0, size = 0; };

class GzFast : public GzStream {
 public:
    // `first` = the stream starts here (an invalid first header is an error; after a member it is the clean end)
    explicit GzFast(int fd, bool first = true, const GzResume *resume = nullptr) : fd_(fd), first_(first) {
        for (auto &c : chunks_) { c.buf.resize(HIST + CHUNK + hulk::inflate::OUT_SLACK + 64); free_.push_back(&c); }
        if (resume) { resume_ = *resume; resuming_ = true; crc_ = resume->crc; size_ = resume->size; }
        th_ = std::thread([this] { produce(); });
    }
    ~GzFast() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
        if (fd_ > 0) ::close(fd_);
    }
    // up to cap bytes; 0 = end of the stream; -1 = error (msg filled)
    long read(uint8_t *dst, size_t cap, std::string &msg) override {
        for (;;) {
            if (!cur_) {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [this] { return !ready_.empty(); });
                cur_ = ready_.front(); ready_.pop_front();
            }
            if (cur_->off < cur_->len) {
                const size_t n = std::min(cap, cur_->len - cur_->off);
                const uint8_t *src = cur_->buf.data() + HIST + cur_->off;
                // (zlib's crc32 takes a uInt length)
                crc_ = hulk::crc32_fast(crc_, src, n);
                memcpy(dst, src, n);
                cur_->off += n; size_ += (uint32_t)n;
                return (long)n;
            }
            // chunk drained: its end-of-member / end-of-stream / error marks
            if (cur_->member_end) {
                if (crc_ != cur_->crc || size_ != cur_->isize) { msg = "gzip: invalid checksum"; return -1; }
                crc_ = 0; size_ = 0;
            }
            if (!cur_->err.empty()) { msg = cur_->err; return -1; }
            const bool eof = cur_->eof;
            if (eof) return 0;                                     // (the chunk stays current: every later call ends here too)
            { std::lock_guard<std::mutex> g(m_); free_.push_back(cur_); }
            cv_.notify_all();
            cur_ = nullptr;
        }
    }

 private:
    static constexpr size_t HIST = 32768, CHUNK = 1u << 20, INBUF = 1u << 20;
    struct Chunk { std::vector<uint8_t> buf; size_t len = 0, off = 0; bool member_end = false, eof = false; uint32_t crc = 0, isize = 0; std::string err; };

    Chunk *get_free() {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [this] { return !free_.empty() || stop_; });
        if (stop_) return nullptr;
        Chunk *c = free_.back(); free_.pop_back();
        c->len = c->off = 0; c->member_end = c->eof = false; c->err.clear();
        return c;
    }
    void publish(Chunk *c) { { std::lock_guard<std::mutex> g(m_); ready_.push_back(c); } cv_.notify_all(); }

    // compressed input: inbuf_[ipos .. iend) is unread (the decoder's `in` pointer is the read position)
    bool refill() {                                                // keeps the unread tail; false = no more input
        const size_t tail = dec_.in ? dec_.in_left() : 0;
        if (tail) memmove(inbuf_.data(), dec_.in, tail);
        size_t got = 0;
        while (!in_eof_ && tail + got < INBUF) {
            const ssize_t m = ::read(fd_, inbuf_.data() + tail + got, INBUF - tail - got);
            if (m < 0) { if (errno == EINTR) continue; io_err_ = std::string("read: ") + strerror(errno); in_eof_ = true; break; }
            if (m == 0) { in_eof_ = true; break; }
            got += (size_t)m;
            break;                                                  // one read per refill is enough
        }
        memset(inbuf_.data() + tail + got, 0, 16);                  // the fast loop loads 8 bytes at a time
        dec_.feed(inbuf_.data(), tail + got);
        return got != 0;
    }
    bool next_byte(uint8_t &b) { while (!dec_.take_byte(b)) { if (!refill()) return false; } return true; }

    // 1 = header read, 0 = clean end (no more members), -1 = error (msg)
    int read_header(bool first, std::string &msg) {
        uint8_t h[10];
        for (int i = 0; i < 10; i++) {
            if (!next_byte(h[i])) {
                if (i == 0 && !first) return 0;
                if (!first && i < 2) return 0;                      // trailing bytes that are not a member
                msg = first && i == 0 ? "EOF" : "unexpected EOF"; return -1;
            }
            if (i == 1 && (h[0] != 0x1f || h[1] != 0x8b)) { if (!first) return 0; msg = "gzip: invalid header"; return -1; }
        }
        if (h[2] != 8) { msg = "gzip: invalid header"; return -1; }
        const uint8_t flg = h[3];
        uint8_t b;
        if (flg & 4) {                                              // FEXTRA
            uint8_t l0, l1; if (!next_byte(l0) || !next_byte(l1)) { msg = "unexpected EOF"; return -1; }
            for (uint32_t n = (uint32_t)l0 | ((uint32_t)l1 << 8); n; n--) if (!next_byte(b)) { msg = "unexpected EOF"; return -1; }
        }
        if (flg & 8) do { if (!next_byte(b)) { msg = "unexpected EOF"; return -1; } } while (b);     // FNAME
        if (flg & 16) do { if (!next_byte(b)) { msg = "unexpected EOF"; return -1; } } while (b);    // FCOMMENT
        if (flg & 2) { if (!next_byte(b) || !next_byte(b)) { msg = "unexpected EOF"; return -1; } }  // FHCRC (not verified)
        return 1;
    }

    // A member always starts a chunk (the end of a member ends its chunk), so the window of the running member is
    // the `hist_have` bytes in front of the chunk's data plus what the chunk holds so far.
    void produce() {
        inbuf_.resize(INBUF + 64);
        dec_.feed(inbuf_.data(), 0);
        Chunk *c = get_free();
        if (!c) return;
        auto finish = [&](Chunk *ch, size_t len, const std::string &e) { ch->len = len; ch->err = e; ch->eof = true; publish(ch); };
        bool first = first_;
        for (;;) {
            uint8_t *base = c->buf.data() + HIST, *out = base;
            size_t hist_have = 0;
            std::string msg;
            if (resuming_) {
                // in the middle of a member: no header; the reader starts `bit` bits into the first byte, the window is handed in
                resuming_ = false; first = false;
                dec_.reset();
                refill();
                dec_.refill_slow();
                if (dec_.bitcnt < resume_.bit) { finish(c, 0, !io_err_.empty() ? io_err_ : std::string("unexpected EOF")); return; }
                dec_.bitbuf >>= resume_.bit; dec_.bitcnt -= resume_.bit;
                hist_have = std::min(HIST, resume_.hist.size());
                memcpy(base - hist_have, resume_.hist.data() + resume_.hist.size() - hist_have, hist_have);
            } else {
                const int hr = read_header(first, msg);
                if (hr <= 0) { finish(c, 0, hr < 0 ? msg : io_err_); return; }
                first = false;
                dec_.reset();
            }
            for (;;) {
                uint8_t *lim = base + CHUNK;
                out = dec_.run(out, lim, base - hist_have, false);
                if (dec_.state == hulk::inflate::Decoder::DONE) break;
                if (dec_.state == hulk::inflate::Decoder::ERROR) { finish(c, (size_t)(out - base), std::string("gzip: ") + dec_.err); return; }
                if (out >= lim) {                                   // chunk full: hand it over, go on in the next one
                    c->len = (size_t)(out - base);
                    Chunk *nx = get_free();
                    if (!nx) return;
                    const size_t hh = std::min(HIST, hist_have + c->len);      // history = tail of (old history + chunk)
                    memcpy(nx->buf.data() + HIST - hh, out - hh, hh);
                    hist_have = hh;
                    publish(c); c = nx;
                    base = c->buf.data() + HIST; out = base;
                    continue;
                }
                if (!refill()) {                                    // starved and nothing more to read
                    out = dec_.run(out, lim, base - hist_have, true);          // reports the truncation
                    if (dec_.state == hulk::inflate::Decoder::DONE) break;
                    if (out >= lim) continue;
                    finish(c, (size_t)(out - base), !io_err_.empty() ? io_err_ : std::string("unexpected EOF"));
                    return;
                }
            }
            // trailer: CRC-32 and ISIZE, little endian, after the next byte boundary
            dec_.align_to_byte();
            uint8_t t[8];
            for (int i = 0; i < 8; i++) if (!next_byte(t[i])) { finish(c, (size_t)(out - base), "unexpected EOF"); return; }
            c->len = (size_t)(out - base);
            c->member_end = true;
            c->crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
            c->isize = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
            Chunk *nx = get_free();
            if (!nx) return;
            publish(c); c = nx;
        }
    }

    int fd_;
    bool first_;
    GzResume resume_;
    bool resuming_ = false;
    std::thread th_;
    std::mutex m_;
    std::condition_variable cv_;
    Chunk chunks_[4];
    std::deque<Chunk *> free_, ready_;
    bool stop_ = false;
    // producer side
    hulk::inflate::Decoder dec_;
    std::vector<uint8_t> inbuf_;
    bool in_eof_ = false;
    std::string io_err_;
    // consumer side
    Chunk *cur_ = nullptr;
    uint32_t crc_ = 0, size_ = 0;
};

// Large scratch buffers of the gzip readers: anonymous mappings that ask for transparent huge pages (a first touch by 16+
// threads at once is otherwise 4 KiB page faults queueing on the process's mapping lock).  HULK_GZ_NO_THP=1: plain pages.
// Regions handed back are kept by the PROCESS for a while (RegionPool): the kernel zeroes pages when they are mapped and, on the
// GPU box, takes as long again to take them back — a run over one 100 MB FASTA file spent 10 ms faulting ~330 MB of block,
// piece and batch buffers in and 16 ms unmapping them, next to 10 ms of parsing (HULK_INGEST_TRACE).  A second file of the
// process finds the regions mapped and touched.  Bounded in size (POOL_BYTES) and in age (FQ_IDLE_SECONDS, swept with the device
// parser's sets: fq_sweep_idle; hulk_release_caches() unmaps at once).  Contents are NOT zeroed on reuse (no user relies on it).
struct RegionPool {
    struct Ent { void *p; size_t n; double t; };
    static constexpr size_t POOL_BYTES = (size_t)1 << 30, ONE_MAX = (size_t)256 << 20;
    std::mutex mu; std::vector<Ent> v; size_t bytes = 0;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    static RegionPool &get() { static RegionPool *g = new RegionPool(); return *g; }      // (never destroyed: buffers of static objects may come back late)
    void *take(size_t n) {                                  // a region of exactly n bytes (sizes are few: block, piece and batch buffers)
        std::lock_guard<std::mutex> g(mu);
        for (size_t i = v.size(); i-- > 0;)
            if (v[i].n == n) { void *p = v[i].p; bytes -= n; v.erase(v.begin() + i); return p; }
        return nullptr;
    }
    bool give(void *p, size_t n) {
        static const bool off = HULK_EXP_ENV("HULK_NO_REGION_POOL") != nullptr;
        if (off || n > ONE_MAX) return false;
        std::lock_guard<std::mutex> g(mu);
        if (bytes + n > POOL_BYTES) return false;
        v.push_back({p, n, now()}); bytes += n;
        return true;
    }
    void sweep(double older_than) {
        std::vector<Ent> drop;
        {
            std::lock_guard<std::mutex> g(mu);
            const double t = now();
            for (size_t i = 0; i < v.size();)
                if (t - v[i].t >= older_than) { drop.push_back(v[i]); bytes -= v[i].n; v.erase(v.begin() + i); } else i++;
        }
        for (auto &e : drop) ::munmap(e.p, e.n);
    }
};
struct BigBuf {
    void *p = nullptr; size_t n = 0;
    BigBuf() {}
    explicit BigBuf(size_t bytes) { reset(bytes); }
    BigBuf(const BigBuf &) = delete;
    BigBuf &operator=(const BigBuf &) = delete;
    ~BigBuf() { release(); }
    void release() { if (p && !RegionPool::get().give(p, n)) ::munmap(p, n); p = nullptr; n = 0; }
    void reset(size_t bytes) {
        release();
        n = (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
        if ((p = RegionPool::get().take(n))) return;
        void *m = ::mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) { p = nullptr; n = 0; throw std::bad_alloc(); }
        p = m;
        static const bool thp = HULK_EXP_ENV("HULK_GZ_NO_THP") == nullptr;
        if (thp) ::madvise(p, n, MADV_HUGEPAGE);
    }
    template <class T> T *as() const { return (T *)p; }
};
// ... with the few members of std::vector<uint8_t> the block reader uses (growing does NOT keep the contents)
struct PageBuf {
    BigBuf m; size_t sz = 0;
    uint8_t *data() const { return m.as<uint8_t>(); }
    size_t size() const { return sz; }
    void resize(size_t n) { if (n > m.n) m.reset(n); sz = n; }
    uint8_t &operator[](size_t i) const { return data()[i]; }
    uint8_t *begin() const { return data(); }
};

// ------------------------------------------------------------------------------------------
// A team of workers that stay around.  run(n, f) calls f(0) .. f(n-1), each once, on the caller and the workers, and returns
// when all are done.  Every parallel step of the ingest path was a fork-join of freshly created threads (two per 32 MB block
// in the parser, three per batch in the gzip readers, one per large copy): hundreds of creations per second of run, each a
// stack mapping under the process's mapping lock, at the moment when dozens of other threads take page faults under the same
// lock.  Indices are handed out one at a time (a task may wait for a LATER index's early result — GzPar's chunks do —, never
// for an earlier one's: the lowest running task can always finish, so fewer awake threads than tasks cannot deadlock).
// ------------------------------------------------------------------------------------------
class Team {
 public:
    explicit Team(unsigned workers) { for (unsigned i = 0; i < workers; i++) th_.emplace_back([this] { loop(); }); }
    ~Team() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    unsigned size() const { return (unsigned)th_.size() + 1; }
    template <class F> void run(unsigned n, F &&f) {
        if (n <= 1 || th_.empty()) { for (unsigned i = 0; i < n; i++) f(i); return; }
        const std::function<void(unsigned)> job(std::ref(f));
        {
            std::lock_guard<std::mutex> g(m_);
            job_ = &job; n_ = n; next_.store(0, std::memory_order_relaxed); pending_ = (unsigned)th_.size(); gen_++;
        }
        cv_.notify_all();
        std::exception_ptr mine;                         // the caller's own share may throw too: the workers still hold `job`
        try {
            for (unsigned i; (i = next_.fetch_add(1, std::memory_order_relaxed)) < n;) f(i);
        } catch (...) {
            mine = std::current_exception();
            next_.store(n, std::memory_order_relaxed);   // nothing more is handed out
        }
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return pending_ == 0; });   // ... and nobody touches `job` or the caller's buffers after this
        if (mine) { failed_ = nullptr; std::rethrow_exception(mine); }
        if (failed_) { std::exception_ptr e = failed_; failed_ = nullptr; std::rethrow_exception(e); }
    }

 private:
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> g(m_);
            cv_.wait(g, [&] { return stop_ || gen_ != seen; });
            if (stop_) return;
            seen = gen_;
            const std::function<void(unsigned)> *job = job_; const unsigned n = n_;
            g.unlock();
            try {
                for (unsigned i; (i = next_.fetch_add(1, std::memory_order_relaxed)) < n;) (*job)(i);
            } catch (...) {                                 // a worker has no caller to unwind to: run() rethrows it in the caller's thread
                next_.store(n, std::memory_order_relaxed);
                g.lock();
                if (!failed_) failed_ = std::current_exception();
                g.unlock();
            }
            g.lock();
            if (--pending_ == 0) done_.notify_one();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    uint64_t gen_ = 0; bool stop_ = false;
    const std::function<void(unsigned)> *job_ = nullptr; unsigned n_ = 0, pending_ = 0;
    std::atomic<unsigned> next_{0};
    std::exception_ptr failed_;
};

// One core copies ~8 GB/s out of a buffer another core wrote, less than several inflate threads deliver: large pieces are
// copied by four threads.  (Pieces of ceil(n / 4) rounded up to 64 bytes: with floor(n / 4), as first written, the last
// n mod 4 bytes were not copied whenever floor(n / 4) happened to be a multiple of 64.)
static void copy_wide(uint8_t *dst, const uint8_t *src, size_t n) {
    if (n < (8u << 20)) { memcpy(dst, src, n); return; }
    const unsigned R = 4;
    const size_t piece = ((n + R - 1) / R + 63) & ~(size_t)63;
    static thread_local Team team(R - 1);                      // (the block reader's thread is the only caller; gone with it)
    team.run(R, [&](unsigned i) { const size_t at = (size_t)i * piece; if (at < n) memcpy(dst + at, src + at, std::min(piece, n - at)); });
}

// ------------------------------------------------------------------------------------------
// BGZF (bgzip / htslib): a gzip file of members of <= 64 KiB of text whose headers carry the member's compressed size in a
// "BC" extra subfield — members are found without inflating and inflated SIDE BY SIDE (one inflate thread is what bounds a
// `.gz` run: 1.2-1.5 GB/s of text).  Purely an optimisation of the same stream semantics: a batch of members is located from
// the BSIZE fields, every member inflated into its own slice of the batch's output (size from its ISIZE trailer) and
// checked — stream ends exactly at the trailer, ISIZE bytes produced, CRC-32 equal.  At the first member that is anything
// else (no BC subfield, other header flags, truncated, any check failing) the members before it are delivered and the
// sequential reader (GzFast) takes the descriptor over from that member's offset: errors, trailing bytes and ordinary
// members keep the exact behaviour and messages of the one-thread path.  Regular files only (needs pread / lseek).
// HULK_GZ_THREADS (default 16, at most the hardware threads; 1 = off).
// ------------------------------------------------------------------------------------------
class GzBgzf : public GzStream {
 public:
    unsigned threads() const { return cfg_.gz_threads; }
    // total size of the member whose header starts at p (n bytes available), 0 = not a BGZF member / header incomplete
    static size_t member_size(const uint8_t *p, size_t n, size_t *header_len) {
        if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || p[3] != 4) return 0;       // FEXTRA and nothing else
        const size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8);
        if (n < 12 + xlen) return 0;
        size_t bsize = 0; bool found = false;
        for (size_t o = 12; o + 4 <= 12 + xlen;) {
            const size_t sl = (size_t)p[o + 2] | ((size_t)p[o + 3] << 8);
            if (o + 4 + sl > 12 + xlen) return 0;
            if (p[o] == 'B' && p[o + 1] == 'C' && sl == 2 && !found) { bsize = (size_t)p[o + 4] | ((size_t)p[o + 5] << 8); found = true; }
            o += 4 + sl;
        }
        if (!found) return 0;
        const size_t total = bsize + 1;
        if (total < 12 + xlen + 2 + 8) return 0;                   // header + the shortest deflate stream + trailer
        *header_len = 12 + xlen;
        return total;
    }
    static bool looks_like(int fd) {
        uint8_t h[64];
        const ssize_t m = ::pread(fd, h, sizeof h, 0);
        size_t hl;
        return m >= 18 && member_size(h, (size_t)m, &hl) != 0;
    }
    GzBgzf(int fd, const IngestCfg &cfg) : cfg_(cfg), fd_(fd) { th_ = std::thread([this] { produce(); }); }
    ~GzBgzf() override {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
        if (cfg_.trace)
            fprintf(stderr, "ingest trace: BGZF reader, %llu members inflated by %u threads%s\n", (unsigned long long)n_members_,
                    threads(), tail_ ? ", then handed over to the sequential reader" : "");
        if (tail_) tail_.reset();                                  // (owns and closes the descriptor from then on)
        else if (fd_ > 0) ::close(fd_);
    }
    long read(uint8_t *dst, size_t cap, std::string &msg) override {
        for (;;) {
            if (tail_) return tail_->read(dst, cap, msg);
            if (!cur_) {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [this] { return !ready_.empty(); });
                cur_ = ready_.front(); ready_.pop_front();
            }
            if (cur_->off < cur_->out_len) {
                const size_t n = std::min(cap, cur_->out_len - cur_->off);
                const uint8_t *src = cur_->out.data() + cur_->off;
                copy_wide(dst, src, n);
                cur_->off += n;
                return (long)n;
            }
            if (!cur_->err.empty()) { msg = cur_->err; return -1; }
            if (cur_->hand_over >= 0) {                            // the rest of the file is the sequential reader's
                if (::lseek(fd_, cur_->hand_over, SEEK_SET) < 0) { msg = std::string("lseek: ") + strerror(errno); return -1; }
                tail_.reset(new GzFast(fd_, !cur_->any_member));
                continue;
            }
            if (cur_->eof) return 0;
            { std::lock_guard<std::mutex> g(m_); free_.push_back(cur_); }
            cv_.notify_all();
            cur_ = nullptr;
        }
    }

 private:
    IngestCfg cfg_;                                          // this run's knobs (resolve_cfg)
    static constexpr size_t IN_BATCH = 8u << 20, MAX_ISIZE = 1u << 16;
    struct Member { size_t hdr_off, in_off, in_len, out_off; uint32_t crc, isize; };
    struct Batch {
        PageBuf in, out; std::vector<Member> mem;      // (huge-page mappings, not zero-filled by one thread: 16 inflate threads touch them first)
        size_t out_len = 0, off = 0; off_t hand_over = -1; bool eof = false, any_member = false; std::string err;
    };
    Batch *get_free() {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [this] { return !free_.empty() || stop_; });
        if (stop_) return nullptr;
        Batch *b = free_.back(); free_.pop_back();
        b->out_len = b->off = 0; b->hand_over = -1; b->eof = false; b->err.clear(); b->mem.clear();
        return b;
    }
    void publish(Batch *b) { { std::lock_guard<std::mutex> g(m_); ready_.push_back(b); } cv_.notify_all(); }
    // one member into its slice; false = anything at all is off (the sequential reader will say what)
    static bool inflate_member(const Batch &b, const Member &m, uint8_t *out) {
        hulk::inflate::Decoder d;
        d.feed(b.in.data() + m.in_off, m.in_len);
        uint8_t *base = out + m.out_off, *end = base + m.isize;
        // one byte of room beyond the slice: the end-of-block code is only read while there is room (a stream that
        // uses it is wrong and is caught by the position check; the byte belongs to a LATER member, which is dropped with it)
        uint8_t *o = d.run(base, end + 1, base, true);
        if (d.state != hulk::inflate::Decoder::DONE || o != end) return false;
        d.align_to_byte();
        if (d.in_left() + (size_t)(d.bitcnt >> 3) != 0) return false;           // the deflate stream ends where BSIZE says
        return hulk::crc32_fast(0, base, m.isize) == m.crc;
    }
    void produce() {
        off_t pos = 0;
        bool any = false;
        for (;;) {
            Batch *b = get_free();
            if (!b) return;
            b->any_member = any;
            b->in.resize(IN_BATCH);
            size_t got = 0;
            while (got < IN_BATCH) {
                const ssize_t r = ::pread(fd_, b->in.data() + got, IN_BATCH - got, pos + (off_t)got);
                if (r < 0) { if (errno == EINTR) continue; b->err = std::string("read: ") + strerror(errno); publish(b); return; }
                if (r == 0) break;
                got += (size_t)r;
            }
            if (got == 0) { b->eof = true; publish(b); return; }
            // the members that lie in this piece of the file, whole
            size_t o = 0, out_total = 0;
            while (o < got) {
                size_t hl = 0;
                const size_t total = member_size(b->in.data() + o, got - o, &hl);
                if (total == 0 || o + total > got) break;
                Member m;
                m.hdr_off = o; m.in_off = o + hl; m.in_len = total - hl - 8; m.out_off = out_total;
                const uint8_t *t = b->in.data() + o + total - 8;
                m.crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
                m.isize = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
                if (m.isize > MAX_ISIZE) break;
                b->mem.push_back(m);
                out_total += m.isize; o += total;
            }
            // (a member cut by the end of the piece starts the next piece; one cut by the end of the FILE, or not a BGZF member, is
            // then the first thing of a piece: the sequential reader's from there on)
            if (b->mem.empty()) { b->hand_over = pos; publish(b); return; }
            if (b->out.size() < out_total + 1) b->out.resize(out_total + out_total / 16 + 1);
            const unsigned T = std::min<unsigned>(threads(), (unsigned)b->mem.size());
            std::atomic<size_t> next{0}, first_bad{b->mem.size()};
            auto work = [&]() {
                for (;;) {
                    const size_t j = next.fetch_add(1);
                    if (j >= b->mem.size() || j > first_bad.load()) return;
                    if (!inflate_member(*b, b->mem[j], b->out.data())) {
                        size_t cur = first_bad.load();
                        while (j < cur && !first_bad.compare_exchange_weak(cur, j)) {}
                    }
                }
            };
            if (!team_) team_.reset(new Team(threads() - 1));
            team_->run(T, [&](unsigned) { work(); });
            const size_t bad = first_bad.load();
            if (bad < b->mem.size()) {
                // everything in front of the first member that failed a check is good; that member starts the hand-over
                const Member &mb = b->mem[bad];
                b->out_len = mb.out_off;
                b->hand_over = pos + (off_t)mb.hdr_off;
                if (bad > 0) any = true;
                n_members_ += bad;
                b->any_member = any;
                publish(b);
                return;
            }
            b->out_len = out_total;
            any = true;
            pos += (off_t)o;
            n_members_ += b->mem.size();
            publish(b);
        }
    }

    int fd_;
    std::thread th_;
    std::mutex m_;
    std::condition_variable cv_;
    Batch batches_[2];
    std::deque<Batch *> free_{&batches_[0], &batches_[1]}, ready_;
    bool stop_ = false;
    Batch *cur_ = nullptr;
    std::unique_ptr<GzFast> tail_;
    uint64_t n_members_ = 0;
    std::unique_ptr<Team> team_;                                  // the producer's
};

// ------------------------------------------------------------------------------------------
// ONE gzip member inflated by several threads (par_inflate.h has the how and why).  The compressed file is taken in batches
// of T chunks of 1 MB; a batch starts at a block boundary `q` that is known to the bit, with the 32 KiB of text in front of it.
// Chunk 0 decodes from q with that window; chunk j > 0 looks for a block header behind byte j MB and decodes from there into
// symbols, not knowing its window; every chunk runs until the first block boundary at or behind the start its successor
// found.  Chunk j's work COUNTS only if chunk j-1 counted and ended, at a block boundary, exactly on chunk j's start bit:
// then its start was a real boundary of the real stream, its unknown symbols are resolved from chunk j-1's last 32 KiB, and
// the member's CRC-32 is combined from the chunks' (zlib's crc32_combine).  Where the chain breaks — a false candidate, no
// candidate, a symbol buffer too small, the end of what was read — the next batch simply starts at the last boundary that
// counted.  Whatever is not the plain middle of a member is not handled here at all: in front of the final block, at a block
// that does not decode, or when a batch made no progress, the bytes so far are delivered and the one-thread reader (GzFast)
// takes the member over from that bit with the window, the CRC-32 and the length so far — so the trailer check, further
// members, trailing bytes, truncation and every message are exactly the one-thread reader's.  Regular files only.
// HULK_GZ_THREADS as for BGZF; HULK_GZ_PAR=0 switches it off; HULK_GZ_PAR_CHUNK (bytes, >= 8 KiB) for tests.
// ------------------------------------------------------------------------------------------
class GzPar : public GzStream {
 public:
    size_t chunk_bytes() const { return cfg_.gz_chunk; }
    // worth its threads and its scratch (2 T symbol buffers of ~21 chunk sizes each) from 4 chunks of compressed input on
    static bool wanted(int fd, const IngestCfg &cfg) {
        struct stat sb;
        return cfg.gz_par && cfg.gz_threads > 1 && fstat(fd, &sb) == 0 && (size_t)sb.st_size >= 4 * cfg.gz_chunk;
    }
    GzPar(int fd, const IngestCfg &cfg) : cfg_(cfg), fd_(fd), t_start_(clock_s()) {
        // no more chunks per batch than the file has: a 5 MB file does not map the scratch of 16 decoders
        struct stat sb;
        if (fstat(fd, &sb) == 0) cfg_.gz_threads = (unsigned)std::max<size_t>(2, std::min<size_t>(cfg_.gz_threads, (size_t)sb.st_size / cfg_.gz_chunk));
        th_ = std::thread([this] { produce(); t_done_ = clock_s() - t_start_; });
    }
    static double clock_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    ~GzPar() override {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        const double t_close = clock_s() - t_start_;
        if (th_.joinable()) th_.join();
        if (cfg_.trace)
            fprintf(stderr, "ingest trace: parallel gzip reader timeline (s since it was opened): first batch out %.3f, last batch out %.3f, producer gone %.3f, "
                    "reader closed %.3f\n", t_first_, t_last_, t_done_, t_close);
        if (cfg_.trace)
            fprintf(stderr, "ingest trace: parallel gzip reader, %llu batches, %llu chunks counted / %llu decoded, %llu bytes of text, %llu members ended here%s; producer s: "
                    "input %.3f, decode %.3f, windows %.3f, waiting for the batch in front to become bytes %.3f\n",
                    (unsigned long long)n_batches_, (unsigned long long)n_counted_, (unsigned long long)n_decoded_, (unsigned long long)n_bytes_,
                    (unsigned long long)n_members_, tail_ ? ", then handed over to the one-thread reader" : "", t_in_, t_dec_, t_win_, t_fin_);
        if (tail_) tail_.reset();                                  // (owns and closes the descriptor from then on)
        else if (fd_ > 0) ::close(fd_);
    }
    long read(uint8_t *dst, size_t cap, std::string &msg) override {
        for (;;) {
            if (tail_) return tail_->read(dst, cap, msg);
            if (!cur_) {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [this] { return !ready_.empty(); });
                cur_ = ready_.front(); ready_.pop_front();
            }
            if (cur_->off < cur_->out_len) {
                const size_t n = std::min(cap, cur_->out_len - cur_->off);
                copy_wide(dst, cur_->out.as<uint8_t>() + cur_->off, n);
                cur_->off += n;
                return (long)n;
            }
            if (!cur_->err.empty()) { msg = cur_->err; return -1; }
            if (cur_->hand) {
                if (::lseek(fd_, cur_->hand_off, SEEK_SET) < 0) { msg = std::string("lseek: ") + strerror(errno); return -1; }
                tail_.reset(cur_->mid_member ? new GzFast(fd_, false, &cur_->resume) : new GzFast(fd_, cur_->stream_start));
                continue;
            }
            { std::lock_guard<std::mutex> g(m_); free_.push_back(cur_); }
            cv_.notify_all();
            cur_ = nullptr;
        }
    }

 private:
    IngestCfg cfg_;                                          // this run's knobs (resolve_cfg)
    static constexpr size_t W = hulk::inflate::SPEC_WINDOW;
    static constexpr int64_t PENDING = -1, NONE = -2;
    struct Batch {
        BigBuf out; size_t out_cap = 0, out_len = 0, off = 0;
        // hand-over to the one-thread reader at byte hand_off: inside a member (resume), or in front of a member's header —
        // the first of the stream (a bad header is an error) or not (it is the clean end)
        bool hand = false, mid_member = false, stream_start = false; off_t hand_off = 0; GzResume resume; std::string err;
    };
    Batch *get_free() {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [this] { return !free_.empty() || stop_; });
        if (stop_) return nullptr;
        Batch *b = free_.back(); free_.pop_back();
        b->out_len = b->off = 0; b->hand = b->mid_member = b->stream_start = false; b->err.clear();
        return b;
    }
    void publish(Batch *b) {
        t_last_ = clock_s() - t_start_; if (t_first_ < 0) t_first_ = t_last_;
        { std::lock_guard<std::mutex> g(m_); ready_.push_back(b); } cv_.notify_all();
    }
    static long pread_all(int fd, uint8_t *dst, size_t len, off_t at) {
        size_t got = 0;
        while (got < len) {
            const ssize_t m = ::pread(fd, dst + got, len - got, at + (off_t)got);
            if (m < 0) { if (errno == EINTR) continue; return -1; }
            if (m == 0) break;
            got += (size_t)m;
        }
        return (long)got;
    }
    // length of the gzip header at p (n bytes there), 0 = not all there / not one
    static size_t header_len(const uint8_t *p, size_t n) {
        if (n < 10 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8) return 0;
        const uint8_t flg = p[3];
        size_t o = 10;
        if (flg & 4) { if (o + 2 > n) return 0; o += 2 + ((size_t)p[o] | ((size_t)p[o + 1] << 8)); if (o > n) return 0; }
        if (flg & 8) { while (o < n && p[o]) o++; if (o >= n) return 0; o++; }
        if (flg & 16) { while (o < n && p[o]) o++; if (o >= n) return 0; o++; }
        if (flg & 2) { o += 2; }
        return o <= n ? o : 0;
    }

    void produce() {
        using namespace hulk::inflate;
        const unsigned T = cfg_.gz_threads;
        const unsigned HW = std::max(1u, std::thread::hardware_concurrency());
        const size_t C = chunk_bytes(), EXTRA = std::max<size_t>(C, 1u << 20), IN_LEN = (size_t)T * C + EXTRA;
        // symbols per chunk: FASTQ deflates 3-5x; a block of zlib's is <= 32 Ki symbols of <= 258 bytes.  Where a chunk does not fit
        // the chain breaks, no more — but a file that keeps breaking it (text that deflates 10x and more) is decoded T times for
        // nothing: after three batches in a row of which less than half counted, the one-thread reader gets the rest
        const size_t CAP = std::max<size_t>(10 * C, 2u << 20);
        unsigned poor = 0;
        // the gzip header
        uint64_t q;                                                // bit of the file where the next batch starts (a block boundary)
        {
            std::vector<uint8_t> h(1u << 16);
            const long m = pread_all(fd_, h.data(), h.size(), 0);
            const size_t hl = m > 0 ? header_len(h.data(), (size_t)m) : 0;
            if (hl == 0) {                                         // the one-thread reader says what is wrong with it
                Batch *b = get_free(); if (!b) return;
                b->hand = true; b->mid_member = false; b->stream_start = true; b->hand_off = 0; publish(b); return;
            }
            q = 8 * (uint64_t)hl;
        }
        BigBuf inbuf[2];
        // Two sets of symbol buffers / tables: while the chunks of batch k+1 are decoded into one, the symbols of batch k are
        // turned into bytes out of the other (all that batch k+1 needs of batch k is where it ended and its last 32 KiB,
        // which the cheap window pass delivers).  Not initialised: pages are touched as far as a chunk gets.
        struct Set { std::vector<BigBuf> sym; std::vector<SpecChunk> ch; std::vector<uint8_t> lut; };
        Set sets[2];
        try {
            for (auto &ib : inbuf) ib.reset(IN_LEN + SPEC_IN_SLACK);
            for (auto &st : sets) {
                st.sym = std::vector<BigBuf>(T); st.ch.resize(T); st.lut.resize((size_t)T * (256 + W));
                for (auto &p : st.sym) p.reset(2 * (W + CAP + SPEC_OUT_SLACK + 8));
            }
        } catch (const std::bad_alloc &) {                       // no room for the scratch (address space, not pages): one thread needs none
            Batch *b = get_free(); if (!b) return;
            b->hand = true; b->mid_member = false; b->stream_start = true; b->hand_off = 0; publish(b); return;
        }
        std::vector<std::atomic<int64_t>> start(T + 1);
        std::vector<uint8_t> win(W, 0);
        size_t have = 0; uint64_t total_ahead = 0;               // (text in front of the batch being decoded)
        uint32_t crc = 0; uint64_t total = 0;                    // of the text handed to the consumer: the finisher's
        // the next batch's input is read ahead while this one is decoded (its position is a guess: this batch's nominal end)
        int cur_in = 0;
        Team decoders(T - 1), resolvers(std::min(HW, 4 * T) - 1);   // (declared in front of the threads that use them: gone after those)
        std::thread ahead; off_t ahead_off = -1; long ahead_got = 0; int ahead_errno = 0;
        std::thread finisher;
        bool gone = false;                                        // the consumer went away while a finisher waited for a free batch
        bool at_final = false, give_up_final = false;
        struct Joiner { std::thread &a, &b; ~Joiner() { if (a.joinable()) a.join(); if (b.joinable()) b.join(); } } joiner{ahead, finisher};
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };

        for (uint64_t k = 0;; k++) {
            Set &S = sets[k & 1];
            // the first batch of a file is a quarter of the others: the parser and the GPU behind this reader start that much sooner
            const unsigned nmax = k == 0 ? std::max(2u, T / 4) : T;
            double t0 = now();
            // input: from the read-ahead when the guess was good, else read now
            if (ahead.joinable()) ahead.join();
            const off_t qb = (off_t)(q >> 3);
            off_t F; long got;
            // (good = the batch starts in the first half of the first chunk of what was read ahead)
            int read_errno = 0;
            if (ahead_off >= 0 && qb >= ahead_off && (size_t)(qb - ahead_off) < std::min(EXTRA, C) / 2) { cur_in ^= 1; F = ahead_off; got = ahead_got; read_errno = ahead_errno; }
            else { F = qb; got = pread_all(fd_, inbuf[cur_in].as<uint8_t>(), IN_LEN, F); read_errno = errno; }
            ahead_off = -1;
            uint8_t *in = inbuf[cur_in].as<uint8_t>();
            if (got >= 0) memset(in + got, 0, SPEC_IN_SLACK);
            const uint64_t in_bits = got > 0 ? 8 * (uint64_t)got : 0, bit0 = q - 8 * (uint64_t)F;
            const bool no_input = got < 0 || in_bits < bit0 + 3;
            if (!no_input && (size_t)got == IN_LEN) {
                ahead_off = F + (off_t)((size_t)nmax * C);
                uint8_t *dst = inbuf[cur_in ^ 1].as<uint8_t>();
                ahead = std::thread([this, dst, IN_LEN, ahead_off, &ahead_got, &ahead_errno] { ahead_got = pread_all(fd_, dst, IN_LEN, ahead_off); ahead_errno = errno; });
            }
            t_in_ += now() - t0; t0 = now();
            if (at_final && !no_input) {
                // The chain arrived in front of the member's final block.  One block, decoded here with the window that is known now;
                // then the trailer is checked against the combined CRC-32 and the length, and if a gzip header follows, the next
                // member is this reader's too.  Anything else about it — the block does not decode, does not end inside what was read, a
                // wrong trailer — and the one-thread reader gets the member from the bit in front of the block, as if this had not been tried.
                at_final = false;
                SpecChunk &c = S.ch[0];
                // (the zero bytes behind the input count as input for the decoder's look-ahead — a final block at the very end of the file
                // has only the 8 trailer bytes behind it —: a stream that really goes on into them fails the position test below)
                c.in = in; c.in_bits = in_bits + 8 * IN_SLACK; c.base = S.sym[0].as<uint16_t>() + W; c.cap = CAP; c.out_len = 0; c.stop = SPEC_ERROR;
                static_assert(IN_SLACK + 8 <= SPEC_IN_SLACK, "the decoder's look-ahead fits the padding behind the input");
                for (size_t i = 0; i < W; i++) c.base[(ptrdiff_t)i - (ptrdiff_t)W] = win[i];
                c.hist_have = have;
                spec_run(c, bit0, [](uint64_t) { return false; }, true);
                const uint64_t tr = (c.end_bit + 7) & ~(uint64_t)7;                 // the trailer, behind the next byte boundary
                bool ok = c.stop == SPEC_LINK && c.blocks == 1 && tr + 64 <= in_bits;
                if (finisher.joinable()) finisher.join();                            // (crc / total are final now)
                if (gone) return;
                Batch *b = nullptr;
                uint32_t crc_all = 0;
                if (ok) {
                    b = get_free();
                    if (!b) return;
                    try {
                        if (b->out_cap < c.out_len) { b->out_cap = 0; b->out.reset(c.out_len + 64); b->out_cap = c.out_len; }
                    } catch (const std::bad_alloc &) {
                        b->out_len = 0; b->err = "gzip: out of memory for " + std::to_string(c.out_len) + " bytes of inflated text";
                        publish(b); return;
                    }
                    uint8_t *l = S.lut.data();
                    for (int i = 0; i < 256; i++) l[i] = (uint8_t)i;
                    memcpy(l + 256, win.data(), W);
                    spec_resolve(c.base, c.out_len, l, b->out.as<uint8_t>());
                    crc_all = (uint32_t)crc32_combine(crc, hulk::crc32_fast(0, b->out.as<uint8_t>(), c.out_len), (z_off_t)c.out_len);
                    const uint8_t *t = in + (tr >> 3);
                    const uint32_t want_crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
                    const uint32_t want_size = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
                    ok = crc_all == want_crc && (uint32_t)(total + c.out_len) == want_size;
                    if (!ok) { std::lock_guard<std::mutex> g(m_); free_.push_back(b); b = nullptr; }
                }
                if (ok) {
                    b->out_len = c.out_len; n_bytes_ += c.out_len; n_members_++;
                    const size_t after = (size_t)(tr >> 3) + 8;                      // first byte behind the trailer, in the buffer
                    const size_t hl = header_len(in + after, (size_t)got - after);
                    // a new member: no window, no checksum, no length
                    crc = 0; total = 0; total_ahead = 0; have = 0;
                    if (hl == 0) {                                                    // no header, or not all of it here: the one-thread reader's
                        b->hand = true; b->mid_member = false; b->stream_start = false; b->hand_off = F + (off_t)after;
                        publish(b);
                        return;
                    }
                    publish(b);
                    q = 8 * ((uint64_t)F + after + hl);
                    continue;
                }
                // (falls through: the ordinary path finds SPEC_FINAL at once, counts nothing, and hands the member over at q)
                give_up_final = true;
            }
            unsigned n = 0, acc = 0;
            std::vector<size_t> off(1, 0);
            uint64_t q_new = q;
            SpecStop last_stop = SPEC_INPUT;
            if (!no_input) {
                n = (unsigned)std::min<size_t>(nmax, ((size_t)got + C - 1) / C);
                for (unsigned j = 0; j <= n; j++) start[j].store(j == 0 ? (int64_t)bit0 : PENDING, std::memory_order_relaxed);
                auto decode = [&](unsigned j) {
                    SpecChunk &c = S.ch[j];
                    c.in = in; c.in_bits = in_bits; c.base = S.sym[j].as<uint16_t>() + W; c.cap = CAP; c.out_len = 0; c.stop = SPEC_ERROR;
                    uint64_t s = bit0;
                    if (j > 0) {
                        const uint64_t from = std::max<uint64_t>(8 * (uint64_t)j * C, bit0 + 1);
                        s = find_block_start(in, in_bits, from, 8 * (uint64_t)j * C + 4 * (uint64_t)C);
                        start[j].store(s == ~0ull ? NONE : (int64_t)s, std::memory_order_release);
                        if (s == ~0ull) return;
                        for (size_t i = 0; i < W; i++) c.base[(ptrdiff_t)i - (ptrdiff_t)W] = (uint16_t)(256 + i);
                        c.hist_have = W;
                    } else {
                        for (size_t i = 0; i < W; i++) c.base[(ptrdiff_t)i - (ptrdiff_t)W] = win[i];
                        c.hist_have = have;
                    }
                    const uint64_t nominal_end = 8 * (uint64_t)(j + 1) * C;
                    spec_run(c, s, [&](uint64_t pos) {
                        if (pos < nominal_end) return false;
                        if (j + 1 >= n) return true;
                        int64_t nx;
                        while ((nx = start[j + 1].load(std::memory_order_acquire)) == PENDING) std::this_thread::yield();
                        return nx == NONE || pos >= (uint64_t)nx;
                    });
                };
                decoders.run(n, decode);
                t_dec_ += now() - t0; t0 = now();
                // the chain: which chunks count
                uint64_t cum = total_ahead;
                for (unsigned j = 0; j < n; j++) {
                    if (j > 0) {
                        const int64_t sj = start[j].load(std::memory_order_relaxed);
                        if (sj < 0 || S.ch[j - 1].stop != SPEC_LINK || S.ch[j - 1].end_bit != (uint64_t)sj || cum < W) break;
                    }
                    acc++; cum += S.ch[j].out_len;
                }
                n_batches_++; n_counted_ += acc; n_decoded_ += n;
                q_new = 8 * (uint64_t)F + S.ch[acc - 1].end_bit;
                last_stop = S.ch[acc - 1].stop;
                // the windows, in order (cheap): table j = identity + the 32 KiB in front of chunk j
                off.assign(acc + 1, 0);
                for (unsigned j = 0; j < acc; j++) {
                    uint8_t *l = S.lut.data() + (size_t)j * (256 + W);
                    for (int i = 0; i < 256; i++) l[i] = (uint8_t)i;
                    memcpy(l + 256, win.data(), W);
                    const uint16_t *e = S.ch[j].base + S.ch[j].out_len;  // (reaches into the window in front of the symbols when the chunk is short)
                    for (size_t i = 0; i < W; i++) win[i] = l[e[(ptrdiff_t)i - (ptrdiff_t)W]];
                    have = std::min(W, have + S.ch[j].out_len);
                    off[j + 1] = off[j] + S.ch[j].out_len;
                }
                total_ahead += off[acc];
                t_win_ += now() - t0; t0 = now();
            }
            const bool stuck = q_new == q;
            q = q_new;
            poor = 2 * acc < n ? poor + 1 : 0;
            // in front of the final block with everything else in order: the next round of this loop tries the member's end
            if (!no_input && last_stop == SPEC_FINAL && !give_up_final && !(poor >= 3)) at_final = true;
            const bool hand = no_input || (last_stop == SPEC_FINAL && !at_final) || last_stop == SPEC_ERROR || (stuck && !at_final) || poor >= 3;
            give_up_final = false;
            // the batch in front has to be out of its symbol buffers (the next decode writes them) and its checksum final
            if (finisher.joinable()) finisher.join();
            t_fin_ += now() - t0;
            if (gone) return;
            GzResume res;
            if (hand) { res.bit = (int)(q & 7); res.hist.assign(win.end() - (ptrdiff_t)have, win.end()); }
            const std::string io_err = got < 0 ? std::string("read: ") + strerror(read_errno) : std::string();
            // bytes and checksums, side by side in pieces of >= 256 KiB, while the next batch is decoded
            finisher = std::thread([this, &S, &crc, &total, &gone, &resolvers, off = std::move(off), acc, hand, res = std::move(res), io_err, q, HW, T]() mutable {
                Batch *b = get_free();
                if (!b) { gone = true; return; }
                const size_t out_total = off[acc];
                struct Piece { unsigned j; size_t at, len; uint32_t crc; };
                std::vector<Piece> pieces;
                // (this thread has no caller to unwind to: running out of memory here must end the run with a message, not the
                //  process with std::terminate — ADVICE r3)
                try {
                    if (b->out_cap < out_total) { b->out_cap = 0; b->out.reset(out_total + out_total / 16 + 64); b->out_cap = out_total + out_total / 16; }
                    pieces.reserve(acc * 8 + 64);
                } catch (const std::bad_alloc &) {
                    b->out_len = 0; b->err = "gzip: out of memory for " + std::to_string(out_total) + " bytes of inflated text";
                    publish(b); gone = true; return;
                }
                const size_t target = std::max<size_t>(256u << 10, out_total / std::max(1u, std::min(HW, 4 * T)) + 1);
                for (unsigned j = 0; j < acc; j++) {
                    const size_t len = S.ch[j].out_len, parts = std::max<size_t>(1, len / target), step = ((len + parts - 1) / parts + 63) & ~(size_t)63;
                    for (size_t at = 0; at < len; at += step) pieces.push_back({j, at, std::min(step, len - at), 0});
                }
                std::atomic<size_t> next{0};
                auto work = [&] {
                    for (size_t i; (i = next.fetch_add(1)) < pieces.size();) {
                        Piece &pc = pieces[i];
                        uint8_t *dst = b->out.as<uint8_t>() + off[pc.j] + pc.at;
                        hulk::inflate::spec_resolve(S.ch[pc.j].base + pc.at, pc.len, S.lut.data() + (size_t)pc.j * (256 + W), dst);
                        pc.crc = hulk::crc32_fast(0, dst, pc.len);
                    }
                };
                resolvers.run((unsigned)std::min<size_t>(pieces.size(), resolvers.size()), [&](unsigned) { work(); });
                for (const Piece &pc : pieces) crc = (uint32_t)crc32_combine(crc, pc.crc, (z_off_t)pc.len);
                total += out_total; n_bytes_ += out_total;
                b->out_len = out_total;
                if (!io_err.empty()) b->err = io_err;
                else if (hand) {
                    b->hand = true; b->mid_member = true; b->hand_off = (off_t)(q >> 3);
                    b->resume = std::move(res); b->resume.crc = crc; b->resume.size = (uint32_t)total;
                }
                publish(b);
            });
            if (hand || !io_err.empty()) return;                   // (the Joiner waits for the finisher)
        }
    }

    int fd_;
    std::thread th_;
    std::mutex m_;
    std::condition_variable cv_;
    Batch batches_[3];
    std::deque<Batch *> free_{&batches_[0], &batches_[1], &batches_[2]}, ready_;
    bool stop_ = false;
    Batch *cur_ = nullptr;
    std::unique_ptr<GzFast> tail_;
    uint64_t n_batches_ = 0, n_counted_ = 0, n_decoded_ = 0, n_bytes_ = 0, n_members_ = 0;
    double t_in_ = 0, t_dec_ = 0, t_win_ = 0, t_fin_ = 0;
    double t_start_ = 0, t_first_ = -1, t_last_ = 0, t_done_ = 0;
};

// ------------------------------------------------------------------------------------------
// Sequential byte source over the inputs.  bufio.Scanner is per input: an unterminated last line is
// a token of THAT input, so a '\n' is supplied at the end of an input that does not end in one.
// ------------------------------------------------------------------------------------------
class ByteSource {
 public:
    ByteSource(const char *const *paths, uint32_t n, const IngestCfg &cfg) : cfg_(cfg) {
        for (uint32_t i = 0; i < n; i++) paths_.push_back(paths[i] ? paths[i] : "");
        stdin_mode_ = paths_.empty();
    }
    ~ByteSource() { close_current(); }

    // up to cap bytes into dst; 0 = all inputs exhausted; -1 = error
    long read(uint8_t *dst, size_t cap, IngestError &err) {
        for (;;) {
            if (!open_) {
                if (stdin_mode_) { if (stdin_done_) return 0; fd_ = 0; open_ = true; is_gz_ = false; }
                else {
                    if (idx_ >= paths_.size()) return 0;
                    if (!open_path(paths_[idx_], err)) return -1;
                }
                last_ = '\n'; got_any_ = false;
            }
            long n;
            if (gzf_) {
                std::string msg;
                n = gzf_->read(dst, cap, msg);
                if (n < 0) { err.set(HULK_ERR_IO, msg); return -1; }
            } else if (is_gz_) {
                n = gzread(gz_, dst, (unsigned)std::min<size_t>(cap, 1u << 30));
                if (n < 0) { int e = 0; const char *m = gzerror(gz_, &e); err.set(HULK_ERR_IO, std::string("gzip: ") + (m ? m : "read error")); return -1; }
            } else if (regular_ && cap >= PAR_READ_MIN && readers() > 1) {
                n = read_pieces(dst, cap);
                if (n < 0) { err.set(HULK_ERR_IO, std::string("read ") + current_name() + ": " + strerror(errno)); return -1; }
            } else {
                do { n = regular_ ? ::pread(fd_, dst, cap, pos_) : ::read(fd_, dst, cap); } while (n < 0 && errno == EINTR);
                if (n < 0) { err.set(HULK_ERR_IO, std::string("read ") + current_name() + ": " + strerror(errno)); return -1; }
                pos_ += (off_t)n;
            }
            if (n > 0) { last_ = dst[n - 1]; got_any_ = true; return n; }
            // end of this input
            const bool need_nl = got_any_ && last_ != '\n';
            close_current();
            if (stdin_mode_) stdin_done_ = true; else idx_++;
            if (need_nl) { dst[0] = '\n'; return 1; }
        }
    }

 private:
    IngestCfg cfg_;                                          // this run's knobs (resolve_cfg)
    // A single read() out of the page cache is one core's memcpy (~10 GB/s), slower than the parser behind
    // it: large requests on a regular file are cut into pieces that are pread() side by side.
    static constexpr size_t PAR_READ_MIN = 8u << 20;
    unsigned readers() const { return cfg_.readers; }
    static long pread_all(int fd, uint8_t *dst, size_t len, off_t at) {
        size_t got = 0;
        while (got < len) {
            const ssize_t m = ::pread(fd, dst + got, len - got, at + (off_t)got);
            if (m < 0) { if (errno == EINTR) continue; return -1; }
            if (m == 0) break;
            got += (size_t)m;
        }
        return (long)got;
    }
    long read_pieces(uint8_t *dst, size_t cap) {
        const unsigned R = readers();
        const size_t piece = (cap / R + 4095) & ~(size_t)4095;
        std::vector<long> got(R, 0);
        std::vector<int> errs(R, 0);
        if (!team_) team_.reset(new Team(R - 1));
        team_->run(R, [&](unsigned i) {
            const size_t at = (size_t)i * piece;
            if (at >= cap) return;
            got[i] = pread_all(fd_, dst + at, std::min(piece, cap - at), pos_ + (off_t)at);
            if (got[i] < 0) errs[i] = errno;
        });
        size_t total = 0;
        for (unsigned i = 0; i < R; i++) {
            if (got[i] < 0) { errno = errs[i]; return -1; }
            total += (size_t)got[i];
            const size_t at = (size_t)i * piece;
            if (at >= cap || (size_t)got[i] < std::min(piece, cap - at)) break;      // end of file inside this piece
        }
        pos_ += (off_t)total;
        return (long)total;
    }
    std::string current_name() const { return stdin_mode_ ? "STDIN" : paths_[idx_]; }
    bool open_path(const std::string &p, IngestError &err) {
        fd_ = ::open(p.c_str(), O_RDONLY);
        if (fd_ < 0) return err.set(HULK_ERR_IO, "open " + p + ": " + strerror(errno));   // os.Open's *PathError text
        open_ = true; pos_ = 0;
        struct stat sb;
        regular_ = fstat(fd_, &sb) == 0 && S_ISREG(sb.st_mode);
        // sketch.go:64-65: strings.Split(name, ".") last element == "gz"
        const size_t dot = p.rfind('.');
        is_gz_ = dot != std::string::npos && p.compare(dot + 1, std::string::npos, "gz") == 0;
        if (is_gz_) {
            uint8_t magic[2] = {0, 0};
            const ssize_t m = ::pread(fd_, magic, 2, 0);
            if (m == 0) { close_current(); return err.set(HULK_ERR_IO, "EOF"); }                    // gzip.NewReader on an empty file
            if (m < 2 || magic[0] != 0x1f || magic[1] != 0x8b) { close_current(); return err.set(HULK_ERR_IO, "gzip: invalid header"); }
            const bool use_zlib = cfg_.zlib;
            if (!use_zlib) {
                if (regular_ && cfg_.gz_threads > 1 && GzBgzf::looks_like(fd_)) gzf_.reset(new GzBgzf(fd_, cfg_));
                else if (regular_ && GzPar::wanted(fd_, cfg_)) gzf_.reset(new GzPar(fd_, cfg_));
                else gzf_.reset(new GzFast(fd_));
                return true;
            }
            gz_ = gzdopen(fd_, "rb");
            if (!gz_) { close_current(); return err.set(HULK_ERR_IO, "gzip: cannot open stream"); }
            gzbuffer(gz_, 1u << 20);
        }
        return true;
    }
    void close_current() {
        if (gzf_) { gzf_.reset(); fd_ = -1; }                     // (the gzip readers close the descriptor)
        else if (gz_) { gzclose(gz_); gz_ = nullptr; fd_ = -1; }  // gzclose closes the descriptor
        else if (fd_ > 0) ::close(fd_);
        fd_ = -1; open_ = false;
    }
    std::vector<std::string> paths_;
    size_t idx_ = 0;
    bool stdin_mode_ = false, stdin_done_ = false, open_ = false, is_gz_ = false, got_any_ = false;
    int fd_ = -1;
    bool regular_ = false;
    off_t pos_ = 0;
    gzFile gz_ = nullptr;
    std::unique_ptr<GzStream> gzf_;
    uint8_t last_ = '\n';
    std::unique_ptr<Team> team_;                                  // read_pieces' readers
};

// ------------------------------------------------------------------------------------------
// Reader thread: blocks that end on '\n' (the unterminated tail is carried into the next block).
// ------------------------------------------------------------------------------------------
struct Block {
    PageBuf buf;
    size_t len = 0;
    bool tail_too_long = false;   // the line after this block's last '\n' already has >= MAX_TOKEN bytes
};

class BlockReader {
 public:
    BlockReader(const char *const *paths, uint32_t n, const IngestCfg &cfg) : block_(cfg.block), src_(paths, n, cfg) { th_ = std::thread([this] { run(); }); }
    ~BlockReader() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
    }
    // next block, or nullptr at the end / on error (err filled)
    std::unique_ptr<Block> next(IngestError &err) {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [this] { return !q_.empty() || done_; });
        if (!q_.empty()) { auto b = std::move(q_.front()); q_.pop_front(); cv_.notify_all(); return b; }
        if (err_.code != HULK_OK) err = err_;
        return nullptr;
    }
    void recycle(std::unique_ptr<Block> b) { std::lock_guard<std::mutex> g(m_); if (pool_.size() < 3) pool_.push_back(std::move(b)); }
    uint64_t bytes_in() const { return bytes_in_; }

 private:
    void run() {
        std::vector<uint8_t> carry;
        bool eof = false;
        while (!eof) {
            std::unique_ptr<Block> b;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [this] { return q_.size() < 2 || stop_; });
                if (stop_) break;
                if (!pool_.empty()) { b = std::move(pool_.back()); pool_.pop_back(); }
            }
            if (!b) b.reset(new Block);
            if (b->buf.size() < block_ + MAX_TOKEN + 16) b->buf.resize(block_ + MAX_TOKEN + 16);
            size_t have = carry.size();
            if (have > b->buf.size() - block_) b->buf.resize(have + block_ + 16);
            if (have) memcpy(b->buf.data(), carry.data(), have);
            carry.clear();
            IngestError e;
            while (have < block_) {
                const long n = src_.read(b->buf.data() + have, block_ - have, e);
                if (n < 0) { finish(e); return; }
                if (n == 0) { eof = true; break; }
                have += (size_t)n; bytes_in_ += (uint64_t)n;
            }
            // cut at the last '\n'
            size_t cut = have;
            while (cut > 0 && b->buf[cut - 1] != '\n') cut--;
            b->tail_too_long = false;
            if (!eof) {
                carry.assign(b->buf.begin() + cut, b->buf.begin() + have);
                if (carry.size() >= MAX_TOKEN) b->tail_too_long = true;
            }   // at EOF every input ended in '\n' (ByteSource), so cut == have
            b->len = eof ? have : cut;
            const bool fatal_tail = b->tail_too_long;
            {
                std::lock_guard<std::mutex> g(m_);
                q_.push_back(std::move(b));
            }
            cv_.notify_all();
            if (fatal_tail) break;       // the parser reports "token too long" after this block
        }
        IngestError none;
        finish(none);
    }
    void finish(const IngestError &e) {
        { std::lock_guard<std::mutex> g(m_); err_ = e; done_ = true; }
        cv_.notify_all();
    }
    const size_t block_;                                     // IngestCfg::block
    ByteSource src_;
    std::thread th_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::unique_ptr<Block>> q_, pool_;
    bool done_ = false, stop_ = false;
    IngestError err_;
    std::atomic<uint64_t> bytes_in_{0};                            // (read by the calling thread for the statistics, also while the reader still runs: an error stops the parser first)
};

// ------------------------------------------------------------------------------------------
// Sinks: where parsed sequences go.  prepare() hands out room for n sequences / nbytes bases
// (lens[i] receives the length of sequence i); commit() takes the first n_commit of them.
// ------------------------------------------------------------------------------------------
struct Sink {
    virtual ~Sink() {}
    virtual bool prepare(uint64_t n, uint64_t nbytes, uint8_t **bases, uint64_t **lens, IngestError &err) = 0;
    virtual bool commit(uint64_t n_commit, IngestError &err) = 0;     // lens -> offsets happens here
    virtual bool finish(IngestError &err) { (void)err; return true; }
    uint64_t n_seqs = 0, total_len = 0;
};

// lens[0..n) -> exclusive offsets in place (array has n+1 entries); returns total, min, max
static uint64_t lens_to_offsets(uint64_t *a, uint64_t n, uint64_t &mn, uint64_t &mx) {
    uint64_t run = 0; mn = ~0ull; mx = 0;
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t L = a[i];
        if (L < mn) mn = L;
        if (L > mx) mx = L;
        a[i] = run; run += L;
    }
    a[n] = run;
    return run;
}

struct CallbackSink : Sink {
    hulk_batch_fn fn; void *user;
    std::vector<uint8_t> bases; std::vector<uint64_t> lens;
    CallbackSink(hulk_batch_fn f, void *u) : fn(f), user(u) {}
    bool prepare(uint64_t n, uint64_t nbytes, uint8_t **b, uint64_t **l, IngestError &) override {
        if (bases.size() < nbytes + 16) bases.resize(nbytes + 16);
        if (lens.size() < n + 2) lens.resize(n + 2);
        *b = bases.data(); *l = lens.data();
        return true;
    }
    bool commit(uint64_t n, IngestError &err) override {
        if (n == 0) return true;
        uint64_t mn, mx;
        const uint64_t tot = lens_to_offsets(lens.data(), n, mn, mx);
        n_seqs += n; total_len += tot;
        if (fn) { const int rc = fn(user, bases.data(), lens.data(), n); if (rc != 0) return err.set(rc < 0 ? rc : HULK_ERR_ARG, "batch callback failed"); }
        return true;
    }
};

#define ING_HIP(call)                                                                               \
    do { hipError_t e_ = (call); if (e_ != hipSuccess) return err.set(HULK_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

// pinned double-buffered staging in front of hulk_add_reads_device
// HULK_INGEST_TRACE=1: seconds the calling thread spent in each phase of a run, on stderr when the run ends (diagnosis)
struct PhaseTrace {
    double wait_block = 0, parse = 0, stage_wait = 0, enqueue = 0, add_reads = 0;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
};

struct GpuSink : Sink {
    // the staging (two pinned + device sets) is the context's: nothing is allocated or freed per run after the first
    hulk_ctx *ctx; hulk::StageSet st{}; uint64_t min_len; PhaseTrace &g_trace;      // (the run's own trace)
    GpuSink(hulk_ctx *c, PhaseTrace &tr) : ctx(c), min_len(hulk::ctx_min_read_len(c)), g_trace(tr) {}
    bool prepare(uint64_t n, uint64_t nbytes, uint8_t **b, uint64_t **l, IngestError &err) override {
        const double tw0 = PhaseTrace::now();
        const int rc = hulk::ctx_stage_acquire(ctx, (size_t)nbytes, n, &st);
        g_trace.stage_wait += PhaseTrace::now() - tw0;
        if (rc != HULK_OK) return err.set(rc, hulk_last_error(ctx));
        *b = st.h_bases; *l = st.h_off;
        return true;
    }
    bool commit(uint64_t n, IngestError &err) override {
        if (n == 0) return true;
        uint64_t mn, mx;
        const uint64_t tot = lens_to_offsets(st.h_off, n, mn, mx);
        // NewMinimizerSketch's checks (minimizer.go:70-76), as hulk_add_reads makes them
        if (mn < 1) return err.set(HULK_ERR_EMPTY_SEQ, hulk_strerror(HULK_ERR_EMPTY_SEQ));
        if (mn < min_len) return err.set(HULK_ERR_SHORT_SEQ, hulk_strerror(HULK_ERR_SHORT_SEQ));
        if (mx > 0xffffffffull) return err.set(HULK_ERR_READ_TOO_LONG, hulk_strerror(HULK_ERR_READ_TOO_LONG));
        hipStream_t stream = hulk::ctx_stream(ctx);
        const double tc0 = PhaseTrace::now();
        ING_HIP(hipMemcpyAsync(st.d_bases, st.h_bases, tot, hipMemcpyHostToDevice, stream));
        ING_HIP(hipMemcpyAsync(st.d_off, st.h_off, (n + 1) * 8, hipMemcpyHostToDevice, stream));
        const double tc1 = PhaseTrace::now(); g_trace.enqueue += tc1 - tc0;
        hulk::ctx_hint_host_offsets(ctx, st.h_off);               // (long sequences: their lengths are read here, not fetched back)
        int rc = hulk_add_reads_device(ctx, st.d_bases, st.d_off, n, (uint32_t)mx, st.cap_bases);
        g_trace.add_reads += PhaseTrace::now() - tc1;
        if (rc != HULK_OK) return err.set(rc, hulk_last_error(ctx));
        rc = hulk::ctx_stage_release(ctx);
        if (rc != HULK_OK) return err.set(rc, hulk_last_error(ctx));
        n_seqs += n; total_len += tot;
        return true;
    }
    bool finish(IngestError &) override { return true; }      // (the sets stay the context's; whoever takes one next waits for its event)
};

// ------------------------------------------------------------------------------------------
// FASTQ: the line machine of FastqHandler.Run.  state = number of filled slots (0..3).
//   state 0..2: an EMPTY line leaves the slot nil (skipped); a non-empty line fills it
//   state 3   : ANY line (empty too) is l4 and completes the record
// ------------------------------------------------------------------------------------------
static inline size_t line_len(const uint8_t *p, const uint8_t *nl) {     // ScanLines' dropCR
    size_t L = (size_t)(nl - p);
    if (L && p[L - 1] == '\r') L--;
    return L;
}

struct Scan1 {
    uint8_t end_state[4]; uint64_t nseq[4], nbytes[4]; uint64_t n_lines = 0; bool too_long = false;
};

static void fastq_pass1(const uint8_t *a, const uint8_t *b, Scan1 &r) {
    uint8_t st[4] = {0, 1, 2, 3};
    uint64_t ns[4] = {0, 0, 0, 0}, nb[4] = {0, 0, 0, 0};
    const uint8_t *p = a;
    while (p < b) {
        const uint8_t *nl = (const uint8_t *)memchr(p, '\n', (size_t)(b - p));
        if (!nl) nl = b;                                   // cannot happen: pieces end in '\n'
        if ((size_t)(nl - p) >= MAX_TOKEN) { r.too_long = true; break; }
        const size_t L = line_len(p, nl);
        r.n_lines++;
        for (int h = 0; h < 4; h++) {
            const uint8_t s = st[h];
            if (s == 3) st[h] = 0;
            else if (L) { if (s == 1) { ns[h]++; nb[h] += L; } st[h] = (uint8_t)(s + 1); }
        }
        p = nl + 1;
    }
    for (int h = 0; h < 4; h++) { r.end_state[h] = st[h]; r.nseq[h] = ns[h]; r.nbytes[h] = nb[h]; }
}

struct Scan2 {
    uint64_t completed = 0;          // records completed in this piece
    bool bad_done = false;           // a record that STARTED here with a bad header completed here
    std::string bad_done_hdr;
    bool bad_pending = false;        // the record in progress at the end started here with a bad header
    std::string bad_pending_hdr;
    bool started = false;            // a header line was seen in this piece
};

static void fastq_pass2(const uint8_t *a, const uint8_t *b, uint8_t state, uint8_t *out, uint64_t *lens, Scan2 &r) {
    const uint8_t *p = a;
    bool cur_bad = false; std::string cur_hdr;
    while (p < b) {
        const uint8_t *nl = (const uint8_t *)memchr(p, '\n', (size_t)(b - p));
        if (!nl) nl = b;
        if ((size_t)(nl - p) >= MAX_TOKEN) break;          // reported from pass 1
        const size_t L = line_len(p, nl);
        if (state == 3) {
            r.completed++;
            if (cur_bad && !r.bad_done) { r.bad_done = true; r.bad_done_hdr = cur_hdr; }
            cur_bad = false; state = 0;
        } else if (L) {
            if (state == 0) {
                r.started = true;
                cur_bad = p[0] != '@';
                if (cur_bad) cur_hdr.assign((const char *)p, std::min<size_t>(L, 512));
            } else if (state == 1) {
                memcpy(out, p, L); out += L; *lens++ = (uint64_t)L;
            }
            state++;
        }
        p = nl + 1;
    }
    if (cur_bad && state != 0) { r.bad_pending = true; r.bad_pending_hdr = cur_hdr; }
}

struct Parser {
    Sink &sink; uint32_t threads; IngestError &err;
    uint64_t n_lines = 0;
    Parser(Sink &s, uint32_t t, IngestError &e) : sink(s), threads(t ? t : 1), err(e) {}

    // ---- FASTQ ----
    uint8_t fq_state = 0;
    std::vector<uint8_t> pending;      // sequence of the record in progress (l2 set, l4 not yet seen)
    bool have_pending = false;
    bool carry_bad = false; std::string carry_hdr;

    bool fastq_block(const Block &blk) { return fastq_bytes(blk.buf.data(), blk.len, blk.tail_too_long); }
    // `len` bytes that end in '\n'; tail_too_long: the unterminated line behind them already has MAX_TOKEN bytes
    bool fastq_bytes(const uint8_t *base, size_t blen, bool tail_too_long) {
        const uint8_t *end = base + blen;
        uint32_t P = (uint32_t)std::min<size_t>(threads, std::max<size_t>(1, blen / 16384));
        std::vector<const uint8_t *> cutp(P + 1);
        cutp[0] = base; cutp[P] = end;
        for (uint32_t i = 1; i < P; i++) {
            const uint8_t *q = base + blen * i / P;
            if (q < cutp[i - 1]) q = cutp[i - 1];
            const uint8_t *nl = q < end ? (const uint8_t *)memchr(q, '\n', (size_t)(end - q)) : nullptr;
            cutp[i] = nl ? nl + 1 : end;
        }
        std::vector<Scan1> s1(P);
        run_parallel(P, [&](uint32_t i) { fastq_pass1(cutp[i], cutp[i + 1], s1[i]); });
        // serial prefix: real start state and output offsets of every piece
        std::vector<uint8_t> st(P + 1);
        std::vector<uint64_t> seq0(P + 1), byte0(P + 1);
        st[0] = fq_state; seq0[0] = have_pending ? 1 : 0; byte0[0] = have_pending ? pending.size() : 0;
        for (uint32_t i = 0; i < P; i++) {
            st[i + 1] = s1[i].end_state[st[i]];
            seq0[i + 1] = seq0[i] + s1[i].nseq[st[i]];
            byte0[i + 1] = byte0[i] + s1[i].nbytes[st[i]];
            n_lines += s1[i].n_lines;
        }
        uint8_t *ob = nullptr; uint64_t *ol = nullptr;
        if (!sink.prepare(seq0[P], byte0[P], &ob, &ol, err)) return false;
        if (have_pending) { memcpy(ob, pending.data(), pending.size()); ol[0] = pending.size(); }
        std::vector<Scan2> s2(P);
        run_parallel(P, [&](uint32_t i) { fastq_pass2(cutp[i], cutp[i + 1], st[i], ob + byte0[i], ol + seq0[i], s2[i]); });
        // errors in stream order (seqio.go:38-40 fires when the record's 4th line arrives)
        for (uint32_t i = 0; i < P; i++) {
            if (carry_bad && s2[i].completed) return bad_id(carry_hdr);
            if (s2[i].bad_done) return bad_id(s2[i].bad_done_hdr);
            if (s2[i].started || s2[i].completed) { carry_bad = s2[i].bad_pending; carry_hdr = s2[i].bad_pending_hdr; }
            if (s1[i].too_long) return err.set(HULK_ERR_LINE_TOO_LONG, hulk_strerror(HULK_ERR_LINE_TOO_LONG));
        }
        if (tail_too_long) return err.set(HULK_ERR_LINE_TOO_LONG, hulk_strerror(HULK_ERR_LINE_TOO_LONG));
        fq_state = st[P];
        // a record whose sequence line has been seen but not its 4th line is not a read yet
        uint64_t n = seq0[P];
        if (fq_state >= 2 && n > 0) {
            const uint64_t L = ol[n - 1];
            uint64_t off = byte0[P] - L;
            pending.assign(ob + off, ob + off + L);
            have_pending = true; n--;
        } else if (fq_state < 2) {
            have_pending = false;
        }
        return sink.commit(n, err);
    }
    bool bad_id(const std::string &hdr) {
        return err.set(HULK_ERR_FASTQ_ID, std::string("read ID in fastq file does not begin with @: ") + hdr);
    }

    // ---- FASTA (sketch.go:102-135: the sequence lines of a '>' record concatenated; an EMPTY line ends the parsing) ----
    // Records are unbounded, lines are not.  A block is cut into pieces at line ends; the pieces are parsed side by side — every
    // piece compacts its sequence lines into a buffer of its own and notes where header lines fell — and copied side by side to
    // the end of `fa_bases`; what is left to do in stream order is a walk over the (few) headers.  (Until round 6: one thread,
    // one std::vector::insert per 60-byte line — 1.6 GB/s of file, 20x below what the long-sequence kernels take.)
    struct RawBuf {                                                     // bytes without a constructor (a vector's resize zero-fills), on huge pages:
        uint8_t *p = nullptr; size_t n = 0, cap = 0;                    // 100 MB of 4 KB pages are 25 k page faults to fill and as many to unmap
        static constexpr size_t FIRST = (size_t)128 << 20;              // a batch (64 MB) + a block + a record's tail fit: growing is the exception,
        ~RawBuf() { if (p && !(cap == FIRST && RegionPool::get().give(p, cap))) ::munmap(p, cap); }   // and the region goes back to the process's pool
        uint8_t *grow(size_t add) {
            if (n + add > cap) {
                const size_t nc = (std::max(std::max(cap * 2, FIRST), n + add + 4096) + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
                void *q = p ? ::mremap(p, cap, nc, MREMAP_MAYMOVE) : (nc == FIRST ? RegionPool::get().take(nc) : nullptr);
                if (!q) {
                    q = ::mmap(nullptr, nc, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
                    if (q != MAP_FAILED) ::madvise(q, nc, MADV_HUGEPAGE);
                } else if (p && q != MAP_FAILED) ::madvise(q, nc, MADV_HUGEPAGE);
                if (q == MAP_FAILED) throw std::bad_alloc();
                p = (uint8_t *)q; cap = nc;
            }
            uint8_t *r = p + n; n += add; return r;
        }
        void erase_front(size_t k) { if (k) { memmove(p, p + k, n - k); n -= k; } }
        size_t size() const { return n; }
    };
    struct FaPiece {
        std::unique_ptr<BigBuf> buf; size_t cap = 0, nbytes = 0;        // the piece's sequence bytes (huge pages; the buffer lives as long as the parser: no page faults per block)
        std::vector<uint64_t> hdr_at;                                   // offsets into buf at which a header line stood
        uint64_t n_lines = 0; bool stopped = false, too_long = false;   // lines seen up to the event; an empty line / a line of >= 64 KiB ends the piece
    };
    bool fa_have_hdr = false, fa_stopped = false;
    std::vector<FaPiece> fa_pieces;
    RawBuf fa_bases; std::vector<uint64_t> fa_lens; uint64_t fa_cur = 0;   // complete records (fa_lens) then the record in progress (fa_cur bytes)

    bool fasta_flush_batch(bool final_record) {
        // everything but the record still being accumulated
        uint64_t n = fa_lens.size(), nbytes = fa_bases.size() - (final_record ? 0 : fa_cur);
        if (n == 0) return true;
        uint8_t *ob; uint64_t *ol;
        if (!sink.prepare(n, nbytes, &ob, &ol, err)) return false;
        if (nbytes >= (8u << 20) && threads > 1) {                      // (one core copies ~10 GB/s)
            const uint32_t T = std::min<uint32_t>(threads, 8);
            const size_t piece = ((nbytes + T - 1) / T + 63) & ~(size_t)63;
            run_parallel(T, [&](uint32_t t) { const size_t at = (size_t)t * piece; if (at < nbytes) memcpy(ob + at, fa_bases.p + at, std::min(piece, (size_t)nbytes - at)); });
        } else if (nbytes) memcpy(ob, fa_bases.p, nbytes);
        memcpy(ol, fa_lens.data(), n * 8);
        if (!sink.commit(n, err)) return false;
        fa_bases.erase_front(nbytes);
        fa_lens.clear();
        return true;
    }
    static void fasta_piece(const uint8_t *p, const uint8_t *end, FaPiece &r) {
        if ((size_t)(end - p) + 1 > r.cap) { r.cap = (size_t)(end - p) + 1 + ((size_t)(end - p) >> 3); r.buf.reset(new BigBuf(r.cap)); }
        r.hdr_at.clear(); r.n_lines = 0; r.stopped = r.too_long = false;
        uint8_t *out = r.buf->as<uint8_t>();
        while (p < end) {
            const uint8_t *nl = (const uint8_t *)memchr(p, '\n', (size_t)(end - p));
            if (!nl) nl = end;
            if ((size_t)(nl - p) >= MAX_TOKEN) { r.too_long = true; break; }
            const size_t L = line_len(p, nl);
            r.n_lines++;
            if (L == 0) { r.stopped = true; break; }                    // sketch.go:103-105: break
            if (p[0] == '>') r.hdr_at.push_back((uint64_t)(out - r.buf->as<uint8_t>()));
            else { memcpy(out, p, L); out += L; }
            p = nl + 1;
        }
        r.nbytes = (size_t)(out - r.buf->as<uint8_t>());
    }
    bool fasta_block(const Block &blk) {
        if (fa_stopped) return true;
        const uint8_t *base = blk.buf.data(), *end = base + blk.len;
        const uint32_t P = (uint32_t)std::min<size_t>(threads, std::max<size_t>(1, blk.len / 65536));
        std::vector<const uint8_t *> cutp(P + 1);
        cutp[0] = base; cutp[P] = end;
        for (uint32_t i = 1; i < P; i++) {
            const uint8_t *q = base + blk.len * i / P;
            if (q < cutp[i - 1]) q = cutp[i - 1];
            const uint8_t *nl = q < end ? (const uint8_t *)memchr(q, '\n', (size_t)(end - q)) : nullptr;
            cutp[i] = nl ? nl + 1 : end;
        }
        if (fa_pieces.size() < P) fa_pieces.resize(P);
        std::vector<FaPiece> &pc = fa_pieces;
        run_parallel(P, [&](uint32_t i) { fasta_piece(cutp[i], cutp[i + 1], pc[i]); });
        // pieces count up to the first event in stream order
        uint32_t used = P; bool stop = false, too_long = false;
        std::vector<size_t> at(P + 1, 0);
        for (uint32_t i = 0; i < P; i++) {
            at[i + 1] = at[i] + pc[i].nbytes;
            n_lines += pc[i].n_lines;
            if (pc[i].stopped || pc[i].too_long) { used = i + 1; stop = pc[i].stopped; too_long = pc[i].too_long; break; }
        }
        const size_t old = fa_bases.size(), add = at[used];
        uint8_t *dst = fa_bases.grow(add);
        run_parallel(used, [&](uint32_t i) { if (pc[i].nbytes) memcpy(dst + at[i], pc[i].buf->as<uint8_t>(), pc[i].nbytes); });
        // the headers, in stream order: a header closes the record in progress (or, the first one, drops what stood in front of it)
        size_t rec_start = old - fa_cur;                                // where the record in progress begins
        for (uint32_t i = 0; i < used; i++)
            for (const uint64_t h : pc[i].hdr_at) {
                const size_t g = old + at[i] + (size_t)h;               // the header stood in front of byte g
                if (fa_have_hdr) fa_lens.push_back((uint64_t)(g - rec_start));   // store the current entry
                rec_start = g;
                fa_have_hdr = true;
            }
        if (!fa_have_hdr) { fa_bases.n = 0; rec_start = 0; }             // sequence lines before any header are dropped (l2 = nil)
        fa_cur = fa_bases.size() - rec_start;
        // bytes in front of the FIRST header of the stream (no record yet owns them) go
        {
            uint64_t owned = fa_cur;
            for (const uint64_t L : fa_lens) owned += L;
            if (fa_bases.size() > owned) fa_bases.erase_front(fa_bases.size() - (size_t)owned);
        }
        if (too_long) return err.set(HULK_ERR_LINE_TOO_LONG, hulk_strerror(HULK_ERR_LINE_TOO_LONG));
        if (stop) { fa_stopped = true; return true; }
        if (blk.tail_too_long) return err.set(HULK_ERR_LINE_TOO_LONG, hulk_strerror(HULK_ERR_LINE_TOO_LONG));
        if (fa_bases.size() - fa_cur >= FASTA_BATCH_BYTES && !fasta_flush_batch(false)) return false;
        return true;
    }
    bool fasta_end() {
        // sketch.go:126-135 flushes the final entry unconditionally; with no header line at all the
        // reference dies on l1[0] = 64 (nil slice) — reported as an error here
        if (!fa_have_hdr) return err.set(HULK_ERR_FASTA_HEADER, hulk_strerror(HULK_ERR_FASTA_HEADER));
        fa_lens.push_back(fa_cur);
        fa_cur = 0;
        return fasta_flush_batch(true);
    }

    template <class F> void run_parallel(uint32_t P, F f) {
        if (P == 1) { f(0); return; }
        if (!team_ || team_->size() < P) team_.reset(new Team(P - 1));
        team_->run(P, [&](unsigned i) { f((uint32_t)i); });
    }
    std::unique_ptr<Team> team_;
};

int run_ingest(const char *const *paths, uint32_t n_paths, int fasta, const IngestCfg &cfg, Sink &sink, PhaseTrace &g_trace,
               hulk_ingest_stats *stats, IngestError &err) {
    const auto t0 = std::chrono::steady_clock::now();
    if (n_paths && !paths) { err.set(HULK_ERR_ARG, "NULL path list"); return err.code; }
    uint32_t threads = cfg.parser_threads;
    // (0: one per hardware thread, at most 16 — 32 and 64 were measured slower on a 256-thread host; a caller's figure is taken as it is)
    if (threads == 0) { threads = std::thread::hardware_concurrency(); if (threads == 0) threads = 1; if (threads > 16) threads = 16; }
    if (threads > 256) threads = 256;
    bool ok = true;
    double t_body_end = 0.0;
    {
        BlockReader reader(paths, n_paths, cfg);
        Parser ps(sink, threads, err);
        for (;;) {
            const double tb0 = PhaseTrace::now();
            std::unique_ptr<Block> b = reader.next(err);
            const double tb1 = PhaseTrace::now(); g_trace.wait_block += tb1 - tb0;
            if (!b) { ok = err.code == HULK_OK; break; }
            ok = fasta ? ps.fasta_block(*b) : ps.fastq_block(*b);
            g_trace.parse += PhaseTrace::now() - tb1;
            reader.recycle(std::move(b));
            if (!ok || (fasta && ps.fa_stopped)) break;
        }
        if (ok && fasta) ok = ps.fasta_end();
        if (ok) ok = sink.finish(err);
        if (stats) {
            stats->n_seqs = sink.n_seqs; stats->total_len = sink.total_len; stats->n_lines = ps.n_lines;
            stats->bytes_in = reader.bytes_in();
            stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        t_body_end = PhaseTrace::now();
    }                                                             // (the reader's thread and blocks, the parser's team and buffers go here)
    if (cfg.trace)
        fprintf(stderr, "ingest trace (calling thread, s): next block %.3f | parse + sink %.3f, of which: staging set (wait / first "
                        "allocation) %.3f, copies queued %.3f, hulk_add_reads_device %.3f | releasing the reader and the parser %.3f\n", g_trace.wait_block, g_trace.parse,
                g_trace.stage_wait, g_trace.enqueue, g_trace.add_reads, PhaseTrace::now() - t_body_end);
    return ok ? HULK_OK : err.code;
}


// ------------------------------------------------------------------------------------------
// FASTQ -> reads on the DEVICE (hulk_fastq.hip).  The host's part of a run shrinks to moving bytes: a reader thread fills
// pinned buffers with raw file bytes (ByteSource: files, gzip, STDIN — fixed-size blocks, cut anywhere), the calling thread
// queues one host-to-device copy and one chain of parse kernels per block and hands the parsed reads of the block before to
// hulk_add_reads_device.  Streams: copies on `cs`, parse kernels on `ps`, binning on the context's lanes;
//   copy(b) -> parse(b) [after parse(b-1): the tail; after the binning of block b-2 has read the output set] -> the block's
//   scalars reach the host -> hulk_add_reads_device(b) while copy(b+1) / parse(b+1) are already under way.
// Buffers: 6 pinned blocks, 4 raw device blocks (porch + block), 3 output sets (bases + offsets), one set of line-index arrays
// — 6 x 16 MiB = 96 MB pinned and about 300 MB of HBM at the default block size.  They belong to the PROCESS, not to a context: a
// run borrows an idle set for its device and block size and hands it back (a `hulk sketch` per file on fresh contexts would
// otherwise pin and unpin 64 MB per file: tens of milliseconds each).  The pool is bounded in size (FQ_POOL_MAX sets) and in AGE:
// a set nobody borrowed for FQ_IDLE_SECONDS is freed by the next hulk_create / hulk_destroy / hulk_sketch_files of the process
// (fq_sweep_idle), so a host that sketched one file does not hold the buffers for as long as it keeps using the library;
// hulk_release_caches() frees them at once.
// ------------------------------------------------------------------------------------------
struct FqDev {
    // FASTQ keeps DEPTH blocks queued behind the one whose reads it hands over (with one, the PCIe link idled a third of the time:
    // a block's copy was queued only when the parse of the block before the previous one had been waited for): a block's raw slot is
    // its successor's tail source and, should the host parser take over, the successor's successor's — NRAW = DEPTH + 2; an output
    // set is written again when DEPTH blocks behind it have been queued — NOUT = DEPTH + 1
    static constexpr int DEPTH = 2, NRAW = DEPTH + 2, NOUT = DEPTH + 1, NHOST = 6, NST = 6;
    int device = 0; size_t block = 0; uint32_t porch = 0;
    double idle_since = 0.0;                                       // when the set went back to the pool (steady clock, seconds)
    hipStream_t cs = nullptr, ps = nullptr;
    uint8_t *d_raw[NRAW] = {}, *h_buf[NHOST] = {}, *d_bases[NOUT] = {};
    uint64_t *d_off[NOUT] = {};
    hulk::FqState *d_state = nullptr, *h_state = nullptr;          // [NST]
    hipEvent_t ev_copied[NHOST] = {}, ev_parsed[NST] = {}, ev_busy[NOUT][2] = {};
    bool busy0[NOUT] = {}, busy1[NOUT] = {};
    hulk::FqBuffers B;
    // --fasta (run_ingest_fasta_device), allocated by the first such run of the set: two line indices with room for (64 KiB + block) / 2
    // lines each, two accumulation buffers (sequence bytes of complete records + the record in progress; they grow with the longest
    // record) and their record offsets — about 2 x 135 MB + 2 x (192 MB + 70 MB) of HBM at the default block size
    struct Fasta {
        bool ready = false;
        hulk::FaBuffers B[2];                                      // index arrays: a block is indexed while the one before it is placed
        hipEvent_t ev_placed[NRAW] = {};                           // the block that used raw slot r has been placed (the slot may be overwritten)
        hulk::FaState *d_state = nullptr, *h_state = nullptr;      // [NST]
        uint8_t *acc[2] = {}; size_t acc_cap[2] = {};
        uint64_t *rec_off[2] = {}; size_t rec_cap = 0;
        hipEvent_t ev_busy[2][2] = {};
        bool busy0[2] = {}, busy1[2] = {};
    } fa;
    size_t raw_bytes() const { return (size_t)porch + block + 64; }
    void release() {
        if (cs) hipStreamSynchronize(cs);
        if (ps) hipStreamSynchronize(ps);
        for (auto &b : fa.B) { hipFree(b.wgcnt); hipFree(b.line_end); hipFree(b.linfo); hipFree(b.ldst); hipFree(b.wghdr); hipFree(b.hrel); hipFree(b.wgbytes); }
        for (auto &e : fa.ev_placed) if (e) hipEventDestroy(e);
        hipFree(fa.d_state); if (fa.h_state) hipHostFree(fa.h_state);
        for (int i = 0; i < 2; i++) { hipFree(fa.acc[i]); hipFree(fa.rec_off[i]); for (auto &e : fa.ev_busy[i]) if (e) hipEventDestroy(e); }
        fa = Fasta{};
        for (auto &p : d_raw) { hipFree(p); p = nullptr; }
        for (auto &p : d_bases) { hipFree(p); p = nullptr; }
        for (auto &p : d_off) { hipFree(p); p = nullptr; }
        for (auto &p : h_buf) { if (p) hipHostFree(p); p = nullptr; }
        hipFree(d_state); d_state = nullptr;
        if (h_state) hipHostFree(h_state); h_state = nullptr;
        hipFree(B.wgcnt); hipFree(B.line_end); hipFree(B.linfo); hipFree(B.wgmap); hipFree(B.wgseq); hipFree(B.src_out);
        hipFree(B.lmap); hipFree(B.wgstate); hipFree(B.wgbytes);
        B = hulk::FqBuffers{};
        for (auto &e : ev_copied) { if (e) hipEventDestroy(e); e = nullptr; }
        for (auto &e : ev_parsed) { if (e) hipEventDestroy(e); e = nullptr; }
        for (auto &pr : ev_busy) for (auto &e : pr) { if (e) hipEventDestroy(e); e = nullptr; }
        if (cs) hipStreamDestroy(cs); if (ps) hipStreamDestroy(ps);
        cs = ps = nullptr;
    }
    static void destroy(void *p) { FqDev *d = (FqDev *)p; d->release(); delete d; }
};

// idle buffer sets of the process (at most FQ_POOL_MAX are kept; the others are freed when their run ends)
static std::mutex g_fq_mu;
static std::vector<FqDev *> g_fq_idle;
constexpr size_t FQ_POOL_MAX = 2;
constexpr double FQ_IDLE_SECONDS = 10.0;
static double fq_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void fq_dev_release(FqDev *d) {
    if (!d) return;
    {
        std::lock_guard<std::mutex> g(g_fq_mu);
        if (g_fq_idle.size() < FQ_POOL_MAX) { d->idle_since = fq_now(); g_fq_idle.push_back(d); return; }
    }
    FqDev::destroy(d);
}
}  // namespace
namespace hulk {
void fq_release_idle() {
    std::vector<FqDev *> drop;
    { std::lock_guard<std::mutex> g(g_fq_mu); drop.swap(g_fq_idle); }
    for (FqDev *d : drop) FqDev::destroy(d);
    RegionPool::get().sweep(0.0);
}
// frees the idle sets nobody has borrowed for FQ_IDLE_SECONDS (called from hulk_create / hulk_destroy / hulk_sketch_files)
void fq_sweep_idle() {
    std::vector<FqDev *> drop;
    {
        std::lock_guard<std::mutex> g(g_fq_mu);
        const double t = fq_now();
        for (size_t i = 0; i < g_fq_idle.size();)
            if (t - g_fq_idle[i]->idle_since > FQ_IDLE_SECONDS) { drop.push_back(g_fq_idle[i]); g_fq_idle.erase(g_fq_idle.begin() + i); }
            else i++;
    }
    for (FqDev *d : drop) FqDev::destroy(d);
    RegionPool::get().sweep(FQ_IDLE_SECONDS);
}
}  // namespace hulk
namespace {
// a parser for blocks of `block` bytes on the context's device: an idle set of the process, or a new one
static FqDev *fq_dev_for(hulk_ctx *ctx, size_t block, IngestError &err) {
    const int device = hulk::ctx_device(ctx);
    {
        std::lock_guard<std::mutex> g(g_fq_mu);
        for (size_t i = 0; i < g_fq_idle.size(); i++)
            if (g_fq_idle[i]->device == device && g_fq_idle[i]->block == block) {
                FqDev *d = g_fq_idle[i]; g_fq_idle.erase(g_fq_idle.begin() + i); return d;
            }
        // (a set of another shape makes room)
        if (g_fq_idle.size() >= FQ_POOL_MAX) { FqDev::destroy(g_fq_idle.front()); g_fq_idle.erase(g_fq_idle.begin()); }
    }
    FqDev *d = new FqDev();
    d->device = device; d->block = block; d->porch = 1u << 20;
#define FQ_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { err.set(HULK_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); FqDev::destroy(d); return nullptr; } } while (0)
    FQ_HIP(hipSetDevice(d->device));
    FQ_HIP(hipStreamCreateWithFlags(&d->cs, hipStreamNonBlocking));
    {   // the parse kernels are short and the calling thread waits for their scalars block by block, while the context's lanes keep
        // the chip full with binning kernels: their workgroups go first when CUs come free
        int lo = 0, hi = 0;
        FQ_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        FQ_HIP(hipStreamCreateWithPriority(&d->ps, hipStreamNonBlocking, hi));
    }
    hulk::FqBuffers &B = d->B;
    B.porch = d->porch;
    B.line_cap = (uint32_t)((d->porch + block) / 8 + 1024);
    B.read_cap = B.line_cap / 2;
    B.bytes_cap = d->porch + block;
    for (auto &p : d->d_raw) { FQ_HIP(hipMalloc((void **)&p, d->raw_bytes())); FQ_HIP(hipMemset(p, '\n', d->raw_bytes())); }
    for (auto &p : d->h_buf) FQ_HIP(hipHostMalloc((void **)&p, block, hipHostMallocDefault));
    for (auto &p : d->d_bases) FQ_HIP(hipMalloc((void **)&p, B.bytes_cap + 64));
    for (auto &p : d->d_off) FQ_HIP(hipMalloc((void **)&p, ((size_t)B.read_cap + 2) * 8));
    FQ_HIP(hipMalloc((void **)&d->d_state, FqDev::NST * sizeof(hulk::FqState)));
    FQ_HIP(hipHostMalloc((void **)&d->h_state, FqDev::NST * sizeof(hulk::FqState), hipHostMallocDefault));
    const size_t nchunk = (d->raw_bytes() + 4095) / 4096 + 8, nlwg = ((size_t)B.line_cap + 255) / 256 + 8;      // (hulk_fastq.hip FQ_T = 256)
    FQ_HIP(hipMalloc((void **)&B.wgcnt, nchunk * 4));
    FQ_HIP(hipMalloc((void **)&B.line_end, (size_t)B.line_cap * 4));
    FQ_HIP(hipMalloc((void **)&B.linfo, (size_t)B.line_cap * 4));
    FQ_HIP(hipMalloc((void **)&B.lmap, (size_t)B.line_cap));
    FQ_HIP(hipMalloc((void **)&B.wgmap, nlwg * 4));
    FQ_HIP(hipMalloc((void **)&B.wgstate, nlwg));
    FQ_HIP(hipMalloc((void **)&B.wgseq, nlwg * 4));
    FQ_HIP(hipMalloc((void **)&B.wgbytes, nlwg * 8));
    FQ_HIP(hipMalloc((void **)&B.src_out, ((size_t)B.read_cap + 2) * 4));
    for (auto &e : d->ev_copied) FQ_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : d->ev_parsed) FQ_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &pr : d->ev_busy) for (auto &e : pr) FQ_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    FQ_HIP(hipDeviceSynchronize());
#undef FQ_HIP
    return d;
}

// reader thread of the device path: raw blocks of exactly `block` bytes (the last one shorter) into the pinned buffers
class RawReader {
 public:
    struct Item { int idx = -1; size_t len = 0; bool eof = false; };
    RawReader(const char *const *paths, uint32_t n, const IngestCfg &cfg, FqDev *dev) : dev_(dev), src_(paths, n, cfg) {
        for (int i = 0; i < FqDev::NHOST; i++) free_.push_back(i);
        th_ = std::thread([this] { run(); });
    }
    ~RawReader() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
    }
    bool next(Item &it, IngestError &err) {                  // false: the stream has ended (or failed: err)
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [this] { return !q_.empty() || done_; });
        if (!q_.empty()) { it = q_.front(); q_.pop_front(); return true; }
        if (err_.code != HULK_OK) err = err_;
        return false;
    }
    void recycle(int idx) { { std::lock_guard<std::mutex> g(m_); free_.push_back(idx); } cv_.notify_all(); }
    uint64_t bytes_in() const { return bytes_in_; }

 private:
    void run() {
        for (;;) {
            int idx;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [this] { return !free_.empty() || stop_; });
                if (stop_) break;
                idx = free_.front(); free_.pop_front();
            }
            size_t have = 0; bool eof = false; IngestError e;
            while (have < dev_->block) {
                const long n = src_.read(dev_->h_buf[idx] + have, dev_->block - have, e);
                if (n < 0) { std::lock_guard<std::mutex> g(m_); err_ = e; done_ = true; cv_.notify_all(); return; }
                if (n == 0) { eof = true; break; }
                have += (size_t)n; bytes_in_ += (uint64_t)n;
            }
            { std::lock_guard<std::mutex> g(m_); Item it; it.idx = idx; it.len = have; it.eof = eof; q_.push_back(it); }
            cv_.notify_all();
            if (eof) break;
        }
        { std::lock_guard<std::mutex> g(m_); done_ = true; }
        cv_.notify_all();
    }
    FqDev *dev_;
    ByteSource src_;
    std::thread th_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Item> q_;
    std::deque<int> free_;
    bool done_ = false, stop_ = false;
    IngestError err_;
    std::atomic<uint64_t> bytes_in_{0};
};

// hulk_sketch_files over the device parser.  `host_took_over` tells the statistics that the host parser finished the stream.
int run_ingest_device(hulk_ctx *ctx, const char *const *paths, uint32_t n_paths, const IngestCfg &cfg_in, PhaseTrace &g_trace,
                      hulk_ingest_stats *stats, IngestError &err) {
    const auto t0 = std::chrono::steady_clock::now();
    if (n_paths && !paths) { err.set(HULK_ERR_ARG, "NULL path list"); return err.code; }
    // The host's whole job is read() into pinned memory and one PCIe copy per block, so its best settings are not the
    // parser's: 16 MiB blocks read in 16 pieces side by side moved 46 GB/s of a page-cached file (1.5e8 reads/s of 150 bp
    // FASTQ: the PCIe link), 4 pieces 33; 8 MiB blocks 28 (profiles/r05_devparse.txt).  A caller's figures are taken as they are
    // (the line index holds (porch + block) / 8 lines in 8 K workgroups of 1 K: blocks of up to 32 MiB).
    IngestCfg cfg = cfg_in;
    if (!cfg.block_set) cfg.block = (size_t)16u << 20;
    if (!cfg.readers_set) { const unsigned hw = std::thread::hardware_concurrency(); cfg.readers = hw ? std::min(16u, hw) : 4u; }
    const size_t block = std::min<size_t>(cfg.block, (size_t)32u << 20);
    FqDev *D = fq_dev_for(ctx, block, err);
    if (!D) return err.code;
    struct Lease { FqDev *d; ~Lease() { hipStreamSynchronize(d->cs); hipStreamSynchronize(d->ps); fq_dev_release(d); } } lease{D};
    (void)lease;                                                 // (declared before the reader: released after its thread has ended)
#define DEV_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { err.set(HULK_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); return err.code; } } while (0)
    DEV_HIP(hipSetDevice(D->device));
    // (busy0 / busy1 of the output sets survive between runs: the binning kernels of the run before — this context's or
    //  another's — may still be reading a set when this run's first parse is queued; its events say when they are done)
    GpuSink sink(ctx, g_trace);                                  // the host parser's way into the context, should it take over
    uint32_t threads = cfg.parser_threads;
    if (threads == 0) { threads = std::thread::hardware_concurrency(); if (threads == 0) threads = 1; if (threads > 16) threads = 16; }
    Parser hp(sink, threads, err);
    const uint64_t min_len = hulk::ctx_min_read_len(ctx);
    uint64_t n_lines = 0, dev_seqs = 0, dev_len = 0;
    bool on_host = false, ok = true;
    std::vector<uint8_t> carry;                                  // host take-over: bytes behind the last '\n' handed to the parser
    std::deque<RawReader::Item> held;                            // blocks whose reads have not been handed over yet (oldest first)
    uint64_t first_held = 0;                                     // number of the block held.front()
    uint32_t last_tail_lines = 0;
    {
        RawReader reader(paths, n_paths, cfg, D);
        // the host parser over one raw block (cut anywhere): everything up to the last '\n', the rest is carried
        auto host_feed = [&](const uint8_t *p, size_t len, bool eof) -> bool {
            carry.insert(carry.end(), p, p + len);
            size_t cut = carry.size();
            while (cut > 0 && carry[cut - 1] != '\n') cut--;
            const bool too_long = !eof && carry.size() - cut >= MAX_TOKEN;
            bool r = true;
            if (cut || too_long) r = hp.fastq_bytes(carry.data(), cut, too_long);
            carry.erase(carry.begin(), carry.begin() + cut);
            return r;
        };
        // the reads of block x (parsed on the device) -> the context; false: failure (err) — or the host takes over (on_host)
        auto consume = [&](uint64_t x) -> bool {
            const int st = (int)(x % FqDev::NST), o = (int)(x % FqDev::NOUT);
            const double tw0 = PhaseTrace::now();
            if (hipEventSynchronize(D->ev_parsed[st]) != hipSuccess) return err.set(HULK_ERR_HIP, "hipEventSynchronize (device FASTQ parser)");
            g_trace.stage_wait += PhaseTrace::now() - tw0;
            const hulk::FqState S = D->h_state[st];
            if (S.need_host) {
                // the stream goes to the host parser from the last record boundary: the previous block's tail (still in its raw
                // buffer on the device), then this block's bytes and everything behind it, out of the pinned buffers
                on_host = true;
                if (hipStreamSynchronize(D->ps) != hipSuccess) return err.set(HULK_ERR_HIP, "hipStreamSynchronize (device FASTQ parser)");
                carry.clear();
                if (x > 0) {
                    const hulk::FqState Pv = D->h_state[(x - 1) % FqDev::NST];
                    carry.resize(Pv.tail_len);
                    if (Pv.tail_len && hipMemcpy(carry.data(), D->d_raw[(x - 1) % FqDev::NRAW] + Pv.tail_start, Pv.tail_len, hipMemcpyDeviceToHost) != hipSuccess)
                        return err.set(HULK_ERR_HIP, "hipMemcpy (tail of the device FASTQ parser)");
                }
                return true;
            }
            const uint64_t n = S.n_seq - S.pending;
            n_lines += S.n_lines - S.tail_lines; last_tail_lines = S.tail_lines;
            if (n == 0) return true;
            // NewMinimizerSketch's checks (minimizer.go:70-76), as hulk_add_reads makes them
            if (S.min_len < 1) return err.set(HULK_ERR_EMPTY_SEQ, hulk_strerror(HULK_ERR_EMPTY_SEQ));
            if (S.min_len < min_len) return err.set(HULK_ERR_SHORT_SEQ, hulk_strerror(HULK_ERR_SHORT_SEQ));
            const double tc1 = PhaseTrace::now();
            int rc = hulk::ctx_wait_event(ctx, D->ev_parsed[st]);
            if (rc == HULK_OK) rc = hulk_add_reads_device(ctx, D->d_bases[o], D->d_off[o], n, S.max_len, D->B.bytes_cap + 64);
            if (rc == HULK_OK) rc = hulk::ctx_record_busy(ctx, D->ev_busy[o][0], D->ev_busy[o][1], &D->busy1[o]);
            g_trace.add_reads += PhaseTrace::now() - tc1;
            if (rc != HULK_OK) return err.set(rc, hulk_last_error(ctx));
            D->busy0[o] = true;
            dev_seqs += n; dev_len += S.seq_bytes - S.pending_len;
            return true;
        };
        uint64_t b = 0;
        for (;;) {
            RawReader::Item it;
            const double tb0 = PhaseTrace::now();
            const bool got = reader.next(it, err);
            const double tb1 = PhaseTrace::now(); g_trace.wait_block += tb1 - tb0;
            if (!got) { ok = err.code == HULK_OK; break; }
            if (on_host) {
                ok = host_feed(D->h_buf[it.idx], it.len, it.eof);
                g_trace.parse += PhaseTrace::now() - tb1;
                reader.recycle(it.idx);
                if (!ok || it.eof) break;
                continue;
            }
            if (it.len == 0) { reader.recycle(it.idx); break; }     // (end of the stream right on a block border)
            const int r = (int)(b % FqDev::NRAW), o = (int)(b % FqDev::NOUT), st = (int)(b % FqDev::NST);
            // raw slot r held block b - NRAW and served block b - NRAW + 1 as the source of its tail
            if (b >= (uint64_t)FqDev::NRAW) DEV_HIP(hipStreamWaitEvent(D->cs, D->ev_parsed[(b - FqDev::NRAW + 1) % FqDev::NST], 0));
            DEV_HIP(hipMemcpyAsync(D->d_raw[r] + D->porch, D->h_buf[it.idx], it.len, hipMemcpyHostToDevice, D->cs));
            DEV_HIP(hipEventRecord(D->ev_copied[it.idx], D->cs));
            DEV_HIP(hipStreamWaitEvent(D->ps, D->ev_copied[it.idx], 0));
            if (D->busy0[o]) { DEV_HIP(hipStreamWaitEvent(D->ps, D->ev_busy[o][0], 0)); D->busy0[o] = false; }
            if (D->busy1[o]) { DEV_HIP(hipStreamWaitEvent(D->ps, D->ev_busy[o][1], 0)); D->busy1[o] = false; }
            DEV_HIP(hulk::launch_fq_parse(D->ps, D->B, b ? D->d_raw[(b - 1) % FqDev::NRAW] : nullptr, b ? D->d_state + (b - 1) % FqDev::NST : nullptr,
                                          D->d_raw[r], D->d_state + st, (uint32_t)it.len, D->d_off[o], D->d_bases[o]));
            DEV_HIP(hipMemcpyAsync(D->h_state + st, D->d_state + st, sizeof(hulk::FqState), hipMemcpyDeviceToHost, D->ps));
            DEV_HIP(hipEventRecord(D->ev_parsed[st], D->ps));
            g_trace.enqueue += PhaseTrace::now() - tb1;
            held.push_back(it);
            if (held.size() > (size_t)FqDev::DEPTH) {               // block b - DEPTH, while the blocks behind it are copied and parsed
                if (!consume(first_held)) { ok = false; break; }
                if (on_host) break;
                reader.recycle(held.front().idx); held.pop_front(); first_held++;
            }
            b++;
            if (it.eof) break;
        }
        while (ok && !on_host && !held.empty()) {                    // the blocks still queued at the end of the stream, in order
            if (!consume(first_held)) { ok = false; break; }
            if (on_host) break;
            reader.recycle(held.front().idx); held.pop_front(); first_held++;
        }
        if (ok && on_host) {
            // the blocks the device had been given but whose reads were not handed over, in order, then the rest of the stream
            bool eof = false;
            while (ok && !held.empty()) {
                const RawReader::Item it = held.front(); held.pop_front();
                ok = host_feed(D->h_buf[it.idx], it.len, it.eof); eof = it.eof;
                reader.recycle(it.idx);
            }
            while (ok && !eof) {
                RawReader::Item it;
                if (!reader.next(it, err)) { ok = err.code == HULK_OK; break; }
                ok = host_feed(D->h_buf[it.idx], it.len, it.eof); eof = it.eof;
                reader.recycle(it.idx);
            }
        }
        if (ok && !on_host) n_lines += last_tail_lines;           // a record in progress at the end of the stream is dropped, its lines were read
        if (ok) ok = sink.finish(err);
        // nothing of this run may still read the pinned blocks or write the output sets when the next run starts
        hipStreamSynchronize(D->cs); hipStreamSynchronize(D->ps);
        if (stats) {
            stats->n_seqs = sink.n_seqs + dev_seqs; stats->total_len = sink.total_len + dev_len; stats->n_lines = n_lines + hp.n_lines;
            stats->bytes_in = reader.bytes_in();
            stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
    }
#undef DEV_HIP
    if (cfg.trace)
        fprintf(stderr, "ingest trace (device FASTQ parser%s; calling thread, s): next block %.3f | copies + parse kernels queued %.3f, "
                        "waiting for a block's scalars %.3f, hulk_add_reads_device %.3f, host parser %.3f\n", on_host ? ", host parser took over" : "",
                g_trace.wait_block, g_trace.enqueue, g_trace.stage_wait, g_trace.add_reads, g_trace.parse);
    return ok ? HULK_OK : err.code;
}


// ------------------------------------------------------------------------------------------
// --fasta -> sequences ON THE DEVICE (hulk_fastq.hip, k_fa_*): sketch.go:102-135 with the host reduced to read() into pinned
// memory, one PCIe copy per block and the bookkeeping of records in stream order.  Unlike FASTQ a block's result depends on
// where the previous block left the accumulation buffer, and the host learns that from the previous block's scalars: the parse
// kernels of block b are queued when block b-1's scalars have arrived (its copy was queued before: the link stays busy, and
// the parse of a block is shorter than its copy).
//   acc[cur]     : [records handed over][complete records][record in progress]; rec_off[cur][r] = where record r begins
//   a batch      : the complete records, handed to hulk_add_reads_device when they hold FASTA_BATCH_BYTES (the host parser's
//                  rule) or the offsets run short; the record in progress then moves to the front of the other buffer
//   events       : an empty line ends the stream (sketch.go:103-105), a line of 64 KiB or more is bufio.Scanner's error — the
//                  first of the two in stream order counts, as in Parser::fasta_block
// ------------------------------------------------------------------------------------------
static bool fa_ensure(FqDev *D, IngestError &err) {
    FqDev::Fasta &F = D->fa;
    if (F.ready) return true;
#define FA_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return err.set(HULK_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
    const uint32_t line_cap = (uint32_t)((MAX_TOKEN + D->block) / 2 + 2);
    const size_t nchunk = (D->raw_bytes() + 4095) / 4096 + 8, nlwg = ((size_t)line_cap + 255) / 256 + 8;     // (hulk_fastq.hip FA_T = 256)
    for (auto &B : F.B) {
        B.porch = D->porch; B.line_cap = line_cap;
        FA_HIP(hipMalloc((void **)&B.wgcnt, nchunk * 4));
        FA_HIP(hipMalloc((void **)&B.line_end, (size_t)line_cap * 4));
        FA_HIP(hipMalloc((void **)&B.linfo, (size_t)line_cap * 4));
        FA_HIP(hipMalloc((void **)&B.ldst, (size_t)line_cap * 4));
        FA_HIP(hipMalloc((void **)&B.hrel, (size_t)line_cap * 4));
        FA_HIP(hipMalloc((void **)&B.wghdr, nlwg * 4));
        FA_HIP(hipMalloc((void **)&B.wgbytes, nlwg * 8));
    }
    FA_HIP(hipMalloc((void **)&F.d_state, FqDev::NST * sizeof(hulk::FaState)));
    FA_HIP(hipHostMalloc((void **)&F.h_state, FqDev::NST * sizeof(hulk::FaState), hipHostMallocDefault));
    F.rec_cap = (size_t)line_cap + ((size_t)1 << 19);
    for (int i = 0; i < 2; i++) {
        F.acc_cap[i] = (size_t)192 << 20;
        FA_HIP(hipMalloc((void **)&F.acc[i], F.acc_cap[i] + 64));
        FA_HIP(hipMalloc((void **)&F.rec_off[i], (F.rec_cap + 2) * 8));
        for (auto &e : F.ev_busy[i]) FA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    for (auto &e : F.ev_placed) FA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    FA_HIP(hipDeviceSynchronize());
#undef FA_HIP
    F.ready = true;
    return true;
}

int run_ingest_fasta_device(hulk_ctx *ctx, const char *const *paths, uint32_t n_paths, const IngestCfg &cfg_in, PhaseTrace &g_trace,
                            hulk_ingest_stats *stats, IngestError &err) {
    const auto t0 = std::chrono::steady_clock::now();
    if (n_paths && !paths) { err.set(HULK_ERR_ARG, "NULL path list"); return err.code; }
    IngestCfg cfg = cfg_in;
    if (!cfg.block_set) cfg.block = (size_t)16u << 20;              // (as the FASTQ path: the host's job is read() and one copy per block)
    if (!cfg.readers_set) { const unsigned hw = std::thread::hardware_concurrency(); cfg.readers = hw ? std::min(16u, hw) : 4u; }
    const size_t block = std::min<size_t>(cfg.block, (size_t)32u << 20);
    FqDev *D = fq_dev_for(ctx, block, err);
    if (!D) return err.code;
    struct Lease { FqDev *d; ~Lease() { hipStreamSynchronize(d->cs); hipStreamSynchronize(d->ps); fq_dev_release(d); } } lease{D};
    (void)lease;
#define DEV_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { err.set(HULK_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); return err.code; } } while (0)
    DEV_HIP(hipSetDevice(D->device));
    if (!fa_ensure(D, err)) return err.code;
    FqDev::Fasta &F = D->fa;
    const uint64_t min_len = hulk::ctx_min_read_len(ctx);
    int cur = 0;
    uint64_t acc_len = 0, open_start = 0;        // bytes in acc[cur]; where the record in progress begins
    uint64_t rec_count = 0, batch_start = 0;     // headers recorded in rec_off[cur] (the last one opened the record in progress); rec_off[cur][0]
    bool have_hdr = false, stopped = false, ok = true;
    uint64_t bmin = ~0ull, bmax = 0;             // over the complete records not handed over yet
    uint64_t n_lines = 0, dev_seqs = 0, bytes_done = 0;
    hipEvent_t last_placed = nullptr;
    std::vector<uint64_t> h_rec;
    // a buffer is written from acc_len on; what the context still reads of it (records handed over by the run before) lies below —
    // except in a buffer taken over EMPTY: wait for its readers first
    auto wait_readers = [&](int b) -> bool {
        if (F.busy0[b]) { if (hipStreamWaitEvent(D->ps, F.ev_busy[b][0], 0) != hipSuccess) return err.set(HULK_ERR_HIP, "hipStreamWaitEvent (FASTA buffers)"); F.busy0[b] = false; }
        if (F.busy1[b]) { if (hipStreamWaitEvent(D->ps, F.ev_busy[b][1], 0) != hipSuccess) return err.set(HULK_ERR_HIP, "hipStreamWaitEvent (FASTA buffers)"); F.busy1[b] = false; }
        return true;
    };
    // room for `need` bytes in acc[b] (contents up to `keep` survive)
    auto ensure_acc = [&](int b, uint64_t need, uint64_t keep) -> bool {
        if (need <= F.acc_cap[b]) return true;
        size_t nc = F.acc_cap[b];
        while (nc < need) nc *= 2;
        uint8_t *q = nullptr;
        if (hipStreamSynchronize(D->ps) != hipSuccess || hipMalloc((void **)&q, nc + 64) != hipSuccess)
            return err.set(HULK_ERR_HIP, "hipMalloc (FASTA accumulation buffer)");
        if (keep && hipMemcpy(q, F.acc[b], keep, hipMemcpyDeviceToDevice) != hipSuccess) { hipFree(q); return err.set(HULK_ERR_HIP, "hipMemcpy (FASTA accumulation buffer)"); }
        // (what the context queued on the old buffer has to be through before it goes: hipFree waits for the device)
        hipFree(F.acc[b]);
        F.acc[b] = q; F.acc_cap[b] = nc; F.busy0[b] = F.busy1[b] = false;
        return true;
    };
    // hand the complete records of acc[cur] to the context; final: the record in progress is complete too (end of the stream)
    auto hand_over = [&](bool final) -> bool {
        uint64_t n = rec_count ? rec_count - 1 : 0;
        if (final && have_hdr) {
            const uint64_t L = acc_len - open_start;
            bmin = std::min(bmin, L); bmax = std::max(bmax, L);
            if (hipMemcpyAsync(F.rec_off[cur] + rec_count, &acc_len, 8, hipMemcpyHostToDevice, D->ps) != hipSuccess || hipStreamSynchronize(D->ps) != hipSuccess)
                return err.set(HULK_ERR_HIP, "hipMemcpyAsync (last FASTA record)");
            n++;
        }
        if (n) {
            // NewMinimizerSketch's checks (minimizer.go:70-76), as hulk_add_reads makes them
            if (bmin < 1) return err.set(HULK_ERR_EMPTY_SEQ, hulk_strerror(HULK_ERR_EMPTY_SEQ));
            if (bmin < min_len) return err.set(HULK_ERR_SHORT_SEQ, hulk_strerror(HULK_ERR_SHORT_SEQ));
            if (bmax > 0xffffffffull) return err.set(HULK_ERR_READ_TOO_LONG, hulk_strerror(HULK_ERR_READ_TOO_LONG));
            const double tc1 = PhaseTrace::now();
            // the records' offsets, for the long-sequence path's descriptors (the parse stream gets there long before the binning would)
            h_rec.resize(n + 1);
            if (hipMemcpyAsync(h_rec.data(), F.rec_off[cur], (n + 1) * 8, hipMemcpyDeviceToHost, D->ps) != hipSuccess || hipStreamSynchronize(D->ps) != hipSuccess)
                return err.set(HULK_ERR_HIP, "hipMemcpyAsync (FASTA record offsets)");
            hulk::ctx_hint_host_offsets(ctx, h_rec.data());
            int rc = last_placed ? hulk::ctx_wait_event(ctx, last_placed) : HULK_OK;
            if (rc == HULK_OK) rc = hulk_add_reads_device(ctx, F.acc[cur], F.rec_off[cur], n, (uint32_t)bmax, F.acc_cap[cur] + 64);
            if (rc == HULK_OK) rc = hulk::ctx_record_busy(ctx, F.ev_busy[cur][0], F.ev_busy[cur][1], &F.busy1[cur]);
            g_trace.add_reads += PhaseTrace::now() - tc1;
            if (rc != HULK_OK) return err.set(rc, hulk_last_error(ctx));
            F.busy0[cur] = true;
            dev_seqs += n; bytes_done += (final ? acc_len : open_start) - batch_start;
        }
        bmin = ~0ull; bmax = 0;
        return true;
    };
    // the record in progress moves to the front of the other buffer; the next blocks are placed there
    auto switch_buffers = [&](uint64_t room) -> bool {
        const int nb = cur ^ 1;
        const uint64_t part = have_hdr ? acc_len - open_start : 0;
        if (!wait_readers(nb)) return false;                         // (records of `nb` handed over two batches ago: it is written from byte 0 now)
        if (!ensure_acc(nb, part + room + 64, 0)) return false;
        if (part && hipMemcpyAsync(F.acc[nb], F.acc[cur] + open_start, part, hipMemcpyDeviceToDevice, D->ps) != hipSuccess)
            return err.set(HULK_ERR_HIP, "hipMemcpyAsync (FASTA record in progress)");
        if (hipMemsetAsync(F.rec_off[nb], 0, 8, D->ps) != hipSuccess) return err.set(HULK_ERR_HIP, "hipMemsetAsync (FASTA record offsets)");
        cur = nb; acc_len = part; open_start = 0; batch_start = 0; rec_count = have_hdr ? 1 : 0;
        return true;
    };
    {
        RawReader reader(paths, n_paths, cfg, D);
        if (!wait_readers(0) || !wait_readers(1)) return err.code;
        std::deque<RawReader::Item> held;        // blocks indexed, not placed yet (one)
        uint64_t b = 0;                          // blocks indexed so far
        // Block x has been indexed: its scalars -> the stream's books, its sequence lines -> the accumulation buffer.
        // false: the run fails (err).  Sets `stopped` at the empty line that ends the parsing.
        auto place = [&](uint64_t x) -> bool {
            const int st = (int)(x % FqDev::NST), r = (int)(x % FqDev::NRAW);
            const double tw0 = PhaseTrace::now();
            if (hipEventSynchronize(D->ev_parsed[st]) != hipSuccess) return err.set(HULK_ERR_HIP, "hipEventSynchronize (device FASTA parser)");
            g_trace.stage_wait += PhaseTrace::now() - tw0;
            const hulk::FaState S = F.h_state[st];
            const bool stop = S.first_empty != hulk::FA_NONE && S.first_empty < S.long_line;
            const bool too_long = S.long_line != hulk::FA_NONE && !stop;
            n_lines += stop ? (uint64_t)S.first_empty + 1 : too_long ? (uint64_t)S.long_line : (uint64_t)S.n_lines;
            // as Parser::fasta_block: the first of the two events in stream order counts; a line too long ends the run with nothing handed over
            if (too_long) return err.set(HULK_ERR_LINE_TOO_LONG, hulk_strerror(HULK_ERR_LINE_TOO_LONG));
            // the offsets could not take this block's headers: a batch ends early
            if (rec_count + S.n_hdr + 2 > F.rec_cap && (!hand_over(false) || !switch_buffers(S.seq_bytes))) return false;
            if (!ensure_acc(cur, acc_len + S.seq_bytes + 64, acc_len)) return false;
            const double tq0 = PhaseTrace::now();
            if (S.seq_bytes || S.n_hdr) {
                const hipError_t e = hulk::launch_fa_place(D->ps, F.B[x & 1], D->d_raw[r], F.d_state + st, F.acc[cur], acc_len, F.rec_off[cur] + rec_count);
                if (e != hipSuccess) return err.set(HULK_ERR_HIP, std::string("launch_fa_place: ") + hipGetErrorString(e));
            }
            if (hipEventRecord(F.ev_placed[r], D->ps) != hipSuccess) return err.set(HULK_ERR_HIP, "hipEventRecord (FASTA block placed)");
            last_placed = F.ev_placed[r];
            g_trace.enqueue += PhaseTrace::now() - tq0;
            if (S.n_hdr) {
                const uint64_t first = acc_len + S.first_hdr, last = acc_len + S.last_hdr;
                if (have_hdr) { const uint64_t L = first - open_start; bmin = std::min(bmin, L); bmax = std::max(bmax, L); }
                else batch_start = first;                               // (sequence lines in front of the stream's first header: no record owns them)
                if (S.n_hdr > 1) { bmin = std::min<uint64_t>(bmin, S.min_len); bmax = std::max<uint64_t>(bmax, S.max_len); }
                rec_count += S.n_hdr; open_start = last; have_hdr = true;
            }
            acc_len += S.seq_bytes;
            if (!have_hdr) acc_len = 0;                                 // (l2 = nil: sequence lines before any header are dropped)
            if (stop) { stopped = true; return true; }
            if (S.tail_len >= MAX_TOKEN) return err.set(HULK_ERR_LINE_TOO_LONG, hulk_strerror(HULK_ERR_LINE_TOO_LONG));
            // a batch is due (the host parser's rule)
            if (rec_count > 1 && open_start - batch_start >= FASTA_BATCH_BYTES && (!hand_over(false) || !switch_buffers(0))) return false;
            return true;
        };
        for (;;) {
            RawReader::Item it;
            const double tb0 = PhaseTrace::now();
            const bool got = reader.next(it, err);
            const double tb1 = PhaseTrace::now(); g_trace.wait_block += tb1 - tb0;
            if (!got) { ok = err.code == HULK_OK; break; }
            if (it.len == 0) { reader.recycle(it.idx); break; }        // (end of the stream right on a block border)
            const int r = (int)(b % FqDev::NRAW), st = (int)(b % FqDev::NST);
            // the copy, and the block's index: neither needs to know where the blocks before left the accumulation buffer
            if (b >= (uint64_t)FqDev::NRAW) DEV_HIP(hipStreamWaitEvent(D->cs, F.ev_placed[r], 0));     // (the block that used this raw slot)
            DEV_HIP(hipMemcpyAsync(D->d_raw[r] + D->porch, D->h_buf[it.idx], it.len, hipMemcpyHostToDevice, D->cs));
            DEV_HIP(hipEventRecord(D->ev_copied[it.idx], D->cs));
            DEV_HIP(hipStreamWaitEvent(D->ps, D->ev_copied[it.idx], 0));
            DEV_HIP(hulk::launch_fa_index(D->ps, F.B[b & 1], b ? D->d_raw[(b - 1) % FqDev::NRAW] : nullptr, b ? F.d_state + (b - 1) % FqDev::NST : nullptr,
                                          D->d_raw[r], F.d_state + st, (uint32_t)it.len));
            DEV_HIP(hipMemcpyAsync(F.h_state + st, F.d_state + st, sizeof(hulk::FaState), hipMemcpyDeviceToHost, D->ps));
            DEV_HIP(hipEventRecord(D->ev_parsed[st], D->ps));
            g_trace.enqueue += PhaseTrace::now() - tb1;
            held.push_back(it);
            b++;
            if (held.size() > 1) {                                      // block b-2 is placed while block b-1 crosses the link and is indexed
                if (!place(b - 2)) { ok = false; break; }
                reader.recycle(held.front().idx); held.pop_front();
                if (stopped) break;
            }
            if (it.eof) break;
        }
        while (ok && !stopped && !held.empty()) {
            if (!place(b - held.size())) ok = false;
            reader.recycle(held.front().idx); held.pop_front();
        }
        if (ok) {
            // sketch.go:126-135 flushes the final entry unconditionally; with no header line at all the reference dies on l1[0] = 64
            if (!have_hdr) ok = err.set(HULK_ERR_FASTA_HEADER, hulk_strerror(HULK_ERR_FASTA_HEADER));
            else ok = hand_over(true);
        }
        hipStreamSynchronize(D->cs); hipStreamSynchronize(D->ps);
        if (stats) {
            stats->n_seqs = dev_seqs; stats->total_len = bytes_done; stats->n_lines = n_lines;
            stats->bytes_in = reader.bytes_in();
            stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
    }
#undef DEV_HIP
    if (cfg.trace)
        fprintf(stderr, "ingest trace (device FASTA parser; calling thread, s): next block %.3f | copies + kernels queued %.3f, "
                        "waiting for a block's scalars %.3f, hulk_add_reads_device %.3f\n",
                g_trace.wait_block, g_trace.enqueue, g_trace.stage_wait, g_trace.add_reads);
    return ok ? HULK_OK : err.code;
}

}  // namespace

extern "C" {

int hulk_parse_files_opts(const char *const *paths, uint32_t n_paths, int fasta, const hulk_ingest_opts *opts, hulk_batch_fn fn,
                          void *user, hulk_ingest_stats *stats, char *errbuf, uint64_t errbuf_len) {
    IngestError err;
    int rc;
    if (const std::string bad = check_opts(opts); !bad.empty()) {
        err.set(HULK_ERR_ARG, bad); rc = err.code;
    } else {
        CallbackSink sink(fn, user);
        PhaseTrace trace;
        rc = run_ingest(paths, n_paths, fasta, resolve_cfg(opts, 0), sink, trace, stats, err);
    }
    if (errbuf && errbuf_len) {
        const std::string &m = rc == HULK_OK ? std::string() : err.msg;
        const size_t n = std::min<size_t>(m.size(), (size_t)errbuf_len - 1);
        memcpy(errbuf, m.data(), n); errbuf[n] = 0;
    }
    return rc;
}

int hulk_parse_files(const char *const *paths, uint32_t n_paths, int fasta, uint32_t threads, hulk_batch_fn fn,
                     void *user, hulk_ingest_stats *stats, char *errbuf, uint64_t errbuf_len) {
    hulk_ingest_opts o; memset(&o, 0, sizeof o);
    o.parser_threads = threads;
    return hulk_parse_files_opts(paths, n_paths, fasta, threads ? &o : nullptr, fn, user, stats, errbuf, errbuf_len);
}

int hulk_sketch_files_opts(hulk_ctx *ctx, const char *const *paths, uint32_t n_paths, int fasta, const hulk_ingest_opts *opts,
                           hulk_ingest_stats *stats) {
    if (!ctx) return HULK_ERR_ARG;
    if (const std::string bad = check_opts(opts); !bad.empty()) return hulk::ctx_fail(ctx, HULK_ERR_ARG, bad.c_str());
    IngestError err;
    int rc;
    {
        PhaseTrace trace;
        const IngestCfg cfg = resolve_cfg(opts, 0);
        if (!fasta && !cfg.host_parser) rc = run_ingest_device(ctx, paths, n_paths, cfg, trace, stats, err);
        else if (fasta && !cfg.host_parser) rc = run_ingest_fasta_device(ctx, paths, n_paths, cfg, trace, stats, err);
        else {
            GpuSink sink(ctx, trace);
            rc = run_ingest(paths, n_paths, fasta, cfg, sink, trace, stats, err);
        }
    }
    if (rc != HULK_OK) return hulk::ctx_fail(ctx, rc, err.msg.c_str());
    return HULK_OK;
}

int hulk_sketch_files(hulk_ctx *ctx, const char *const *paths, uint32_t n_paths, int fasta, uint32_t threads,
                      hulk_ingest_stats *stats) {
    hulk_ingest_opts o; memset(&o, 0, sizeof o);
    o.parser_threads = threads;
    return hulk_sketch_files_opts(ctx, paths, n_paths, fasta, threads ? &o : nullptr, stats);
}

}  // extern "C"
