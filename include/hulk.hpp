// hulk.hpp — C++ host side above the C ABI (hulk_hip.h): the reference's Go objects for the sketch
// path, with the reference's method names, argument meaning and error texts.
//
//   reference (Go)                                                        here
//   -------------------------------------------------------------------   ---------------------------
//   pipeline.Info / SketchCmd          src/pipeline/pipeline.go           hulk::SketchInfo
//   findMinimizers(chan, *Info)        src/pipeline/boss.go:54            hulk::Boss::FindMinimizers
//   theBoss.AddSeq / Flush / StopWork / GetMinimizerCount   boss.go:24-41 same names on hulk::Boss
//   histosketch.HistoSketch (exported fields)  histosketch.go:36-47       hulk::HistoSketch
//   DataStreamer + FastqHandler + AddSeq loop  pipeline/sketch.go:40-217  hulk::Boss::SketchFiles
//   log.Fatalf("ERROR---> %v") via helpers.ErrorCheck   helpers.go:31-35  hulk::Error (what() = %v)
//   SeqMinimizer.Run's loop with the read stream sharded over GPUs        hulk::Boss::Shard + AddSeq + StopWorkSharded
//       pipeline/sketch.go:182-250                                          (RCCL inside libhulkhip.so)
//
// Header only; link with -lhulkhip.  A Boss is single-caller, like SeqMinimizer.Run's goroutine.
#ifndef HULK_HPP
#define HULK_HPP

#include <algorithm>
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "hulk_hip.h"

namespace hulk {

class Error : public std::runtime_error {
 public:
    Error(int code, const std::string &msg) : std::runtime_error(msg), code_(code) {}
    int code() const { return code_; }
 private:
    int code_;
};

// the flags of `hulk sketch` that reach the path (cmd/sketch.go:50-59, cmd/root.go:62)
struct SketchInfo {
    unsigned KmerSize = 21;        // -k
    unsigned WindowSize = 9;       // -w
    unsigned SketchSize = 50;      // -s
    unsigned Interval = 0;         // -i
    double DecayRatio = 1.0;       // -x
    int32_t SpectrumSize = 0;      // 0 = Pow(KmerSize, 4)  (cmd/sketch.go:118)
    int Device = 0;
    unsigned Rank = 0, World = 1;  // multi-GPU: this process is rank Rank of World (one GPU each); it owns the sketch
                                   // slots [S*Rank/World, S*(Rank+1)/World) and Boss::Shard connects it to the others
};
using UniqueId = std::array<uint8_t, HULK_UNIQUE_ID_BYTES>;

// histosketch.HistoSketch as sketchio consumes it (exported fields only)
struct HistoSketch {
    unsigned KmerSize = 0;
    std::vector<uint64_t> Sketch;          // `mins`
    std::vector<double> SketchWeights;     // `weights`
    unsigned SketchSize = 0;               // `num`
    int32_t Dimensions = 0;                // `num_histogram_bins`
    bool ApplyConceptDrift = false;        // `concept_drift` (decayRatio != 1.0, histosketch.go:79-81)
};

struct IngestStats { uint64_t SeqCount = 0, LengthTotal = 0, Lines = 0, BytesIn = 0; double Seconds = 0; };

class Boss {
 public:
    // findMinimizers + NewHistoSketch: throws hulk::Error with the reference's message
    static Boss FindMinimizers(const SketchInfo &info) { return Boss(info); }

    Boss(Boss &&o) noexcept : ctx_(o.ctx_), info_(o.info_), bins_(o.bins_), sharded_(o.sharded_), bases_(std::move(o.bases_)),
                              offsets_(std::move(o.offsets_)) { o.ctx_ = nullptr; }
    Boss(const Boss &) = delete;
    Boss &operator=(const Boss &) = delete;
    ~Boss() { if (ctx_) hulk_destroy(ctx_); }

    // ---- the read stream sharded over World GPUs (include/hulk_hip.h, hulk_step_sharded): rank 0 draws an id
    // (CommUniqueId), the host hands it to the other ranks, every rank calls Shard — then AddSeq takes THIS RANK's reads:
    // of every step of World * T sketching intervals of the global stream (T = hulk_batch_size) the whole intervals
    // [Rank * T, (Rank + 1) * T), in stream order.  A full share (T * Interval reads) is pushed as one step.
    static UniqueId CommUniqueId() {
        UniqueId id{};
        const int rc = hulk_comm_unique_id(id.data());
        if (rc != HULK_OK) throw Error(rc, hulk_last_error(nullptr));
        return id;
    }
    void Shard(const UniqueId &id) {
        if (info_.Interval == 0) throw Error(HULK_ERR_ARG, "a sharded run needs Interval > 0");
        check(hulk_comm_init(ctx_, id.data(), info_.Rank, info_.World));
        sharded_ = true;
    }
    // End of the stream: `lastStepIntervals` = sketching intervals of the GLOBAL stream in the last, ragged step (the same
    // value on every rank, 0 if the stream ended on a step border); this rank's remaining reads are its share of it.
    void StopWorkSharded(uint32_t lastStepIntervals) {
        if (lastStepIntervals) push_step(lastStepIntervals);
        check(hulk_finish(ctx_));
    }

    // theBoss.AddSeq (boss.go:24-26); sequences are staged and cross the ABI in batches
    void AddSeq(const uint8_t *seq, size_t len) {
        bases_.insert(bases_.end(), seq, seq + len);
        offsets_.push_back(bases_.size());
        if (sharded_) {
            if (offsets_.size() - 1 == (size_t)hulk_batch_size(ctx_) * info_.Interval) push_step(info_.World * hulk_batch_size(ctx_));
        } else if (offsets_.size() > kBatchReads || bases_.size() > kBatchBytes) push();
    }
    void AddSeq(const std::string &seq) { AddSeq(reinterpret_cast<const uint8_t *>(seq.data()), seq.size()); }

    // theBoss.Flush (boss.go:34-36).  With Interval set the library applies the rule of sketch.go:211-215
    // itself; an explicit Flush is only meaningful when Interval == 0.
    void Flush() { push(); check(hulk_flush(ctx_)); }

    // final Flush + theBoss.StopWork (sketch.go:219-224)
    void StopWork() { push(); check(hulk_finish(ctx_)); }

    // theBoss.GetMinimizerCount (boss.go:39-41)
    uint64_t GetMinimizerCount() {
        uint64_t r = 0, m = 0, l = 0;
        check(hulk_get_counters(ctx_, &r, &m, &l));
        return m;
    }

    // DataStreamer.Run + FastqHandler.Run + the AddSeq loop, natively (paths empty = STDIN)
    IngestStats SketchFiles(const std::vector<std::string> &paths, bool fasta = false, unsigned threads = 0) {
        push();
        std::vector<const char *> p;
        for (const auto &s : paths) p.push_back(s.c_str());
        hulk_ingest_stats st{};
        check(hulk_sketch_files(ctx_, p.empty() ? nullptr : p.data(), (uint32_t)p.size(), fasta ? 1 : 0, threads, &st));
        IngestStats out;
        out.SeqCount = st.n_seqs; out.LengthTotal = st.total_len; out.Lines = st.n_lines; out.BytesIn = st.bytes_in;
        out.Seconds = st.seconds;
        return out;
    }

    // the Sketcher's HistoSketch (sketch.go:271-301), valid after StopWork
    HistoSketch Sketch() {
        HistoSketch hs;
        hs.KmerSize = info_.KmerSize; hs.SketchSize = info_.SketchSize; hs.Dimensions = bins_;
        hs.ApplyConceptDrift = info_.DecayRatio != 1.0;
        hs.Sketch.resize(info_.SketchSize); hs.SketchWeights.resize(info_.SketchSize);
        check(sharded_ ? hulk_gather_sketch(ctx_, hs.Sketch.data(), hs.SketchWeights.data())
                       : hulk_get_sketch(ctx_, hs.Sketch.data(), hs.SketchWeights.data()));
        return hs;
    }

    hulk_ctx *handle() { return ctx_; }

 private:
    static constexpr size_t kBatchReads = 1u << 16, kBatchBytes = 64u << 20;
    explicit Boss(const SketchInfo &info) : info_(info) {
        hulk_params p{};
        p.k = info.KmerSize; p.w = info.WindowSize; p.sketch_size = info.SketchSize; p.num_bins = info.SpectrumSize;
        p.decay_ratio = info.DecayRatio; p.interval = info.Interval; p.device = info.Device;
        if (info.World > 1) {
            p.slot_begin = (uint32_t)((uint64_t)info.SketchSize * info.Rank / info.World);
            p.slot_count = (uint32_t)((uint64_t)info.SketchSize * (info.Rank + 1) / info.World) - p.slot_begin;
        }
        const int rc = hulk_create(&p, &ctx_);
        if (rc != HULK_OK) throw Error(rc, hulk_last_error(nullptr));
        bins_ = info.SpectrumSize;
        if (bins_ == 0) { uint64_t b = 1; for (int i = 0; i < 4; i++) b *= info.KmerSize; bins_ = (int32_t)b; }
        offsets_.push_back(0);
    }
    void push() {
        const uint64_t n = offsets_.size() - 1;
        if (n == 0) return;
        const int rc = hulk_add_reads(ctx_, bases_.data(), offsets_.data(), n);
        bases_.clear(); offsets_.assign(1, 0);
        check(rc);
    }
    void push_step(uint32_t stepIntervals) {
        const uint64_t n = offsets_.size() - 1;
        const int rc = hulk_step_sharded_host(ctx_, bases_.data(), offsets_.data(), n, stepIntervals);
        bases_.clear(); offsets_.assign(1, 0);
        check(rc);
    }
    void check(int rc) { if (rc != HULK_OK) throw Error(rc, hulk_last_error(ctx_)); }

    hulk_ctx *ctx_ = nullptr;
    SketchInfo info_;
    int32_t bins_ = 0;
    bool sharded_ = false;
    std::vector<uint8_t> bases_;
    std::vector<uint64_t> offsets_;
};

// sketchio's pairwise distances over loaded sketches (cmd/smash.go:183-226): distances[s*N+q]
inline std::vector<double> Smash(const std::vector<HistoSketch> &sketches, const std::string &metric, int device = 0) {
    const uint32_t N = (uint32_t)sketches.size(), S = N ? sketches[0].SketchSize : 0;
    std::vector<uint64_t> mins((size_t)N * S);
    std::vector<double> weights((size_t)N * S), out((size_t)N * N);
    for (uint32_t i = 0; i < N; i++) {
        if (sketches[i].Sketch.size() != S) throw Error(HULK_ERR_ARG, "sketch length mismatch");
        for (uint32_t j = 0; j < S; j++) { mins[(size_t)i * S + j] = sketches[i].Sketch[j]; weights[(size_t)i * S + j] = sketches[i].SketchWeights[j]; }
    }
    int m;
    if (metric == "jaccard") m = HULK_METRIC_JACCARD;
    else if (metric == "weightedjaccard") m = HULK_METRIC_WEIGHTED_JACCARD;
    else throw Error(HULK_ERR_ARG, "supplied distance metric is not available: " + metric);
    const int rc = hulk_smash(device, mins.data(), weights.data(), N, S, m, out.data());
    if (rc != HULK_OK) throw Error(rc, hulk_strerror(rc));
    return out;
}

// `hulk smash` as the reference runs it (cmd/smash.go:160-226): the sketch files of a directory in, <outFile>.hulk-matrix.csv out —
// LoadHULKdata (JSON, class / version, MD5) for every file, FindSketch(kSize, algo), the matrix on the GPU, the CSV — all inside the
// library (hulk_smash_files).  Returns the distances in sorted-path order, [s * N + q]; throws hulk::Error with the reference's message.
struct SmashStats { double SecondsLoad = 0, SecondsMatrix = 0, SecondsCSV = 0, KernelMs = 0; uint32_t Sketches = 0, SketchSize = 0; };
inline std::vector<double> SmashFiles(const std::vector<std::string> &jsonFiles, uint32_t kSize, const std::string &algo,
                                      const std::string &metric, const std::string &matrixCSV, const std::string &bannerCSV = std::string(),
                                      SmashStats *stats = nullptr, int device = 0, uint32_t threads = 0) {
    std::vector<const char *> ptr;
    for (const auto &f : jsonFiles) ptr.push_back(f.c_str());
    std::vector<std::string> uniq(jsonFiles);
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    std::vector<double> out(uniq.size() * uniq.size());
    hulk_smash_stats st;
    char err[4096] = {0};
    const int rc = hulk_smash_files(device, ptr.data(), (uint32_t)ptr.size(), kSize, algo.c_str(), metric.c_str(), threads,
                                    matrixCSV.empty() ? nullptr : matrixCSV.c_str(), bannerCSV.empty() ? nullptr : bannerCSV.c_str(),
                                    out.data(), &st, err, sizeof err);
    if (rc != HULK_OK) throw Error(rc, err);
    if (stats) *stats = SmashStats{st.seconds_load, st.seconds_matrix, st.seconds_csv, st.kernel_ms, st.n_sketches, st.sketch_size};
    return out;
}

}  // namespace hulk
#endif
