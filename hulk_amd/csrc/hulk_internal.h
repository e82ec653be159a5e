// Internal declarations shared by the kernel file and the C-ABI file of libhulkhip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

// Experiment switches (HULK_JUMP_*, HULK_NIB_*, HULK_NO_FMIN, HULK_K1_DEBUG, HULK_POISON, the HULK_NO_* / HULK_BATCH overrides
// of hulk_params fields ... docs/EXPERIMENTS.md) exist only in the profiling build, `make EXPERIMENTS=1` ->
// libhulkhip_exp.so (tools/ select it with HULK_LIB).  The shipping library reads the environment in two places only:
// HULK_RCCL_LIB (which RCCL to bind) and the ingest overrides of hulk_ingest_opts (hulk_ingest.hip, once per run).
#ifdef HULK_EXPERIMENTS
#define HULK_EXP_ENV(name) getenv(name)
#else
#define HULK_EXP_ENV(name) ((const char *)nullptr)
#endif

namespace hulk {

constexpr int CMS_DEPTH_MAX = 8;      // est[] rows are padded to 8 per bin
constexpr int SCAN_TILE = 1024;       // bins per CWS scan tile (256 threads x float4)
constexpr int SCAN_ROWS = 8;          // sketch slots per CWS scan workgroup
constexpr int SCAN_BATCH_MAX = 16;     // max sketching intervals flushed by one pass over the K table
constexpr int RING_MAX = SCAN_BATCH_MAX + 1;  // spectra in the ring (one may be a partial interval)
constexpr int MIN_SLOTS = 8192;       // max blocks of k_minimizer_bin = per-block minimizer-count slots

// Device-resident run state (one per context).
struct DevState {
    unsigned long long total_len;     // SeqMinimizer.Run lengthTotal     (pipeline/sketch.go:208)
    unsigned long long n_elements;    // AddElement calls (non-zero bins streamed)
    unsigned int used[2][RING_MAX];   // KmerSpectrum.Cardinality() per ring spectrum (ping-pong by flush parity)
    int err;                          // first deferred HULK_ERR_* (0 = none)
    unsigned int skip_exact[2];       // by flush parity: no element of this batch can lower any slot's weight
                                      // (k_flush_decide) -> the estimates, the K scan and the resolve are not run
    unsigned int pad;
};

// One flush = `count` consecutive spectra of the ring starting at ring_base.
struct FlushBatch {
    uint32_t ring_base, ring_n, count;
    int parity;
    int32_t num_bins;
    int prio = 0;       // wave priority (s_setprio) the latency-bound count-min replay kernels raise themselves to
};

// Minimizer list written by k_minimizer_fast: one region of `rcap` entries per wave (16 reads).
constexpr int FAST_READS_PER_WAVE = 16;
constexpr int JUMP_LO_CAP = 64;        // unfinished chains a region may hand to k_jump_left (one round of a wave)
struct MinimizerList {
    uint64_t *x;        // [regions][rcap] distinct minimizer values
    uint8_t *slot;      // [regions][rcap] spectrum (ring slot) of the read each value came from
    uint32_t *key;      // dense (regions back to back): slot << 20 | bin, written by k_jump_bin
    uint32_t *cnt;      // [regions]
    uint32_t *dmask;    // [regions] bit i: read i of the region is deferred to the generic kernel (N, too long, too repetitive)
    uint32_t *dsum;     // [regions / 1024 + 1] per-block sums of popcount(dmask) for the deferred-read list
    uint32_t *off;      // [regions + 1] exclusive prefix of cnt
    uint32_t *bsum;     // [regions / 1024 + 1] per-block sums for the prefix
    uint32_t *partial;  // [max_parts][ring_n][num_bins] per-part spectra of k_range_hist
    uint32_t max_parts;
    uint4 *lo;          // [regions][JUMP_LO_CAP] unfinished jump chains of k_jump_bin: {key lo, key hi, float-as-int t, idx | slot << 16}
    uint32_t *lo_cnt;   // [regions]
    uint32_t *nib;      // [nib_parts][ring_n][nranges][NIB_WORDS] per-part spectra of k_nibble_hist, 8 four-bit counters per word
    uint32_t *nib_over; // [RING_MAX] a 4-bit counter overflowed in this spectrum: k_range_hist recounts it
    uint32_t nib_parts;
    uint64_t rcap;
};

// one long sequence of a group launch (k_long_hash / k_long_emit)
struct LongSeqDesc {
    uint64_t seq_off;    // first base, offset into the bases buffer
    uint64_t L;          // bases
    uint64_t xs_off;     // slice of the per-position scratch (hashed k-mers, valid flags)
    uint64_t tab_off;    // slice of the set table
    uint64_t tab_mask;   // slice size - 1 (power of two)
    uint32_t hslot;      // spectrum (ring slot) of this sequence
    uint32_t pad;
};

struct MinimizerParams {
    uint32_t k, w;
    int32_t num_bins;
    uint32_t xcap;       // max k-mer positions per read this launch supports
    uint32_t tab_size;   // per-wave dedupe table entries (power of two, > xcap)
    uint32_t lds_per_wave;  // filled by launch_minimizer_bin
    uint32_t debug;         // ablation switches (env HULK_K1_DEBUG), 0 in production
    uint32_t skip_long;     // reads beyond xcap are handled by the long-read path, not an error
    uint32_t pair;          // k_minimizer_fast: two 16-lane groups per read (reads of up to 2*16w - (w-1) positions)
    uint64_t bases_bytes;
    uint64_t interval;   // reads per k-mer spectrum (0 = everything into ring_base)
    uint64_t fill;       // reads already counted into the first spectrum of this launch
    uint32_t ring_base, ring_n;   // spectra live in a ring of ring_n histograms
};

// bytes of dynamic LDS per wave / per workgroup for k_minimizer_bin
size_t minimizer_lds_per_wave(uint32_t xcap, uint32_t tab_size);
size_t minimizer_lds_per_block(uint32_t xcap, uint32_t tab_size, int waves);

hipError_t launch_minimizer_bin(hipStream_t s, const uint8_t *d_bases, const uint64_t *d_offsets,
                                uint64_t n_reads, MinimizerParams P, int block_threads,
                                uint32_t *d_hist, DevState *d_state, unsigned long long *d_min_slots,
                                const uint32_t *d_read_list, const uint32_t *d_read_list_count,
                                uint32_t list_blocks);
hipError_t launch_minimizer_fast(hipStream_t s, const uint8_t *d_bases, const uint64_t *d_offsets,
                                 uint64_t n_reads, MinimizerParams P, const MinimizerList &ml,
                                 DevState *d_state, unsigned long long *d_min_slots);
hipError_t launch_minimizer_post(hipStream_t s, uint64_t n_reads, MinimizerParams P, const MinimizerList &ml,
                                 uint32_t *d_hists, uint32_t *d_slow_list, uint32_t *d_slow_count,
                                 hipEvent_t jump_begin = nullptr, hipEvent_t jump_end = nullptr,
                                 hipEvent_t wait_before_spectra = nullptr, hipEvent_t left_begin = nullptr,
                                 hipEvent_t left_end = nullptr);
uint32_t minimizer_list_rcap(uint32_t w, bool pair);
hipError_t launch_long_group(hipStream_t s, const uint8_t *d_bases, const LongSeqDesc *d_desc, uint32_t n_seqs,
                             uint64_t max_npos, MinimizerParams P, uint64_t *d_xs, uint8_t *d_valid, uint64_t *d_table,
                             uint64_t table_total, uint32_t *d_hists, unsigned long long *d_min_slots);
hipError_t launch_count_used(hipStream_t s, const uint32_t *d_hists, DevState *st, const FlushBatch &fb);
hipError_t launch_build_chains(hipStream_t s, uint16_t *d_pos16, uint8_t *d_meta8, int32_t num_bins, int depth, int width);
hipError_t launch_cms_binorder(hipStream_t s, uint32_t *d_hists, const uint16_t *d_pos16, const uint8_t *d_meta8,
                               unsigned long long *d_ctr, uint32_t *d_segsum, unsigned long long *d_base,
                               double *d_f64, float *d_rcp32, int depth, int width, size_t row_stride,
                               DevState *st, const FlushBatch &fb, bool chain = false);
int lds_order_verified(int device);   // 1 / 0 / -(hipError_t): see hulk_countmin.hip
size_t cms_binorder_entries(int depth, int width);   // entries of segsum / base per spectrum
hipError_t launch_cmsd_binorder(hipStream_t s, uint32_t *d_hists, const uint16_t *d_pos16, const uint8_t *d_meta8,
                                const uint32_t *d_eidx, const uint32_t *d_etot, double *d_ctrd, double *d_segadd,
                                double *d_segfac, uint32_t *d_sege0, double *d_cstart, double *d_f64, float *d_rcp32,
                                int depth, int width, size_t row_stride, double omega, DevState *st, const FlushBatch &fb,
                                hipEvent_t freq_begin = nullptr, hipEvent_t freq_end = nullptr, bool chain_form = false);
hipError_t launch_cws_scan(hipStream_t s, const float *d_k32, const float *d_rcp32, float *d_tilemin,
                           int slots, int ntiles, size_t row_stride, DevState *st, const FlushBatch &fb,
                           const float *d_kmin32, float *d_rext, const double *d_weights, int slot_begin,
                           unsigned long long *d_visited, double drift_dw, unsigned long long *d_scanmap, bool per_interval,
                           float *d_rmm, uint32_t *d_scanlist, uint32_t *d_scanlist_n, hipEvent_t scan_begin = nullptr,
                           hipEvent_t scan_end = nullptr);
hipError_t launch_tile_kmin(hipStream_t s, const float *d_k32, float *d_kmin32, int slots, int ntiles, size_t row_stride);
hipError_t launch_slot_kmin(hipStream_t s, const float *d_kmin32, float *d_kminslot, int slots, int ntiles);
hipError_t launch_flush_decide(hipStream_t s, const unsigned long long *d_ctr, int ncounters, const float *d_kminslot,
                               const double *d_weights, int slots, int slot_begin, DevState *st, const FlushBatch &fb,
                               int enable, unsigned long long *d_seal = nullptr, uint32_t seal_tag = 0);
// hulk_step_sharded's exchange (hulk_countmin.hip): a rank's block is SHARD_HDR header words — the 64-bit SEAL
// {word 0: need_full verdict for the next step, word 1: step + 1}, written last and with one store by k_flush_decide, then the
// used bins per interval — and, for the delta exchange, [T][depth * width] count-min increments
constexpr int SHARD_HDR = 32;
constexpr int SHARD_VERDICT = 0, SHARD_TAG = 1, SHARD_USED = 2;
// every rank's block of the gathered header must carry this step's tag (else DevState.err = HULK_ERR_COMM); the gathered
// header is stored to `h_out` (mapped pinned host memory) by the kernel itself, for the host's choice of the next exchange
hipError_t launch_shard_check(hipStream_t s, const uint32_t *d_hdr_all, uint32_t world, uint32_t step_tag, DevState *st,
                              uint32_t *h_out, int fatal);
hipError_t launch_shard_local(hipStream_t s, uint32_t *d_hists, const uint16_t *d_pos16, uint32_t *d_hdr, uint32_t *d_delta,
                              int depth, int width, const FlushBatch &fb);
hipError_t launch_shard_apply(hipStream_t s, const uint32_t *d_hdr_all, const uint32_t *d_delta_all, unsigned long long *d_ctr,
                              int depth, int width, uint32_t world, uint32_t T, uint32_t step_intervals, int32_t num_bins,
                              DevState *st, uint32_t step_tag);
hipError_t launch_cws_resolve(hipStream_t s, const double *d_rcb, const double *d_f64,
                              const float *d_tilemin, double *d_candA, int32_t *d_candB,
                              unsigned long long *d_mins, double *d_weights,
                              int slots, int slot_begin, int ntiles, const unsigned long long *d_scanmap, DevState *st,
                              const FlushBatch &fb);
hipError_t launch_elem_index(hipStream_t s, const uint32_t *d_hists, uint32_t *d_blkcnt, uint32_t *d_eidx,
                             uint32_t *d_etot, const FlushBatch &fb, DevState *st);
int elem_index_blocks(int32_t num_bins);
hipError_t launch_cws_resolve_drift(hipStream_t s, const double *d_rcb, const double *d_f64,
                                    const float *d_tilemin, unsigned long long *d_mins, double *d_weights,
                                    int slots, int slot_begin, int ntiles, double decay_weight, float *d_slotmin,
                                    const unsigned long long *d_scanmap, DevState *st, const FlushBatch &fb);
hipError_t launch_cws_chunk(hipStream_t s, const uint64_t *d_pairs, uint64_t n_attempts, double *d_val,
                            uint32_t *d_blkcnt, unsigned long long *d_gamma_total, unsigned long long *d_chunk_base,
                            double *d_rcb, uint64_t num_bins, uint64_t slot_begin, uint64_t slots,
                            uint64_t sketch_size, double ainv, double bbb, double ccc, double magic,
                            const uint64_t *d_raw, uint64_t first_attempt, const uint64_t *d_ev, uint32_t n_ev);
hipError_t launch_alfg(hipStream_t s, const uint64_t *d_coef, const uint64_t *d_coef_far, uint32_t far_chunks,
                       uint64_t *d_windows, uint64_t *d_raw, uint32_t first_chunk, uint32_t n_chunks, uint64_t chunk_len);
hipError_t launch_rng_candidates(hipStream_t s, const uint64_t *d_raw, uint64_t n, uint64_t *d_list, uint32_t cap,
                                 unsigned int *d_count);
hipError_t launch_cws_beta(hipStream_t s, const uint64_t *d_uraw, uint64_t first_entry, uint64_t n, double *d_rcb,
                           uint64_t num_bins, uint64_t slot_begin, uint64_t slots);
hipError_t launch_smash(hipStream_t s, const unsigned long long *d_mins, const double *d_weights, uint32_t N, uint32_t S,
                        int metric, double *d_out, double *d_mT, double *d_wT);     // d_mT, d_wT: scratch [S][smash_padded_n(N)]
uint32_t smash_padded_n(uint32_t N);
hipError_t launch_build_k32(hipStream_t s, const double *d_rcb, float *d_k32, int slots,
                            int32_t num_bins, size_t row_stride);
hipError_t launch_selftest_rcp(hipStream_t s, unsigned long long *d_mismatches);
hipError_t launch_fill_f32(hipStream_t s, float *p, size_t n, float v);
hipError_t launch_add_hist(hipStream_t s, uint32_t *d_hist, const uint32_t *d_add, int32_t num_bins);

// Per-kernel timing of WHOLE launch chains (hulk_set_profiling bit 32 -> hulk_get_profile_table): every launch site of the step
// path names the kernel it is about to launch; while a context with that bit set is driving the launches on this thread
// (ProfScope, hulk_flush.hip) an event is recorded on the stream in front of it, and a kernel's duration is the time to the
// next mark on the same stream ("-" closes a chain).  Meant for the one-stream mode (HULK_FLAG_NO_OVERLAP): every kernel alone.
void prof_mark(hipStream_t s, const char *kernel);

// context accessors for hulk_ingest.hip (defined in hulk_flush.hip; not part of the ABI)
}  // namespace hulk
struct hulk_ctx;
namespace hulk {
hipStream_t ctx_stream(hulk_ctx *c);
uint64_t ctx_min_read_len(const hulk_ctx *c);
int ctx_fail(hulk_ctx *c, int code, const char *full_message);
// The next of the context's two pinned + device staging sets (the ones hulk_add_reads stages host buffers through), grown to
// `nbytes` of bases and `n` reads; returns once the copies and kernels that last used it are done.  They belong to the
// context and live until hulk_destroy: a run of hulk_sketch_files neither allocates nor frees pinned memory after the first
// (freeing two 19 MB pinned buffers was 20 ms of a 70 ms run over 2·10^6 reads).  ctx_stage_release: call after the copies
// and kernels reading the set are queued on the context's stream.
struct StageSet { uint8_t *h_bases, *d_bases; uint64_t *h_off, *d_off; size_t cap_bases; };
int ctx_stage_acquire(hulk_ctx *c, size_t nbytes, uint64_t n, StageSet *out);
int ctx_stage_release(hulk_ctx *c);
// for the device FASTQ parser of hulk_sketch_files (hulk_ingest.hip / hulk_fastq.hip)
int ctx_device(const hulk_ctx *c);
void fq_release_idle();      // hulk_release_caches: the idle buffer sets of the process (hulk_ingest.hip)
void fq_sweep_idle();        // ... those idle for more than 10 s only (hulk_create / hulk_destroy / hulk_sketch_files)
// the context's stream waits for `e` (a parse that filled device buffers the next hulk_add_reads_device reads)
int ctx_wait_event(hulk_ctx *c, hipEvent_t e);
// record e0 on the context's stream and, if there is a second work lane, e1 on it: both passed = the kernels queued so far
// have read their inputs.  *has1 says whether e1 was recorded.
int ctx_record_busy(hulk_ctx *c, hipEvent_t e0, hipEvent_t e1, bool *has1);
// offsets[0 .. n] of the NEXT hulk_add_reads_device as the caller holds them in host memory (read during that call only): the
// long-sequence path then takes the lengths from there instead of fetching them from the device behind everything queued
void ctx_hint_host_offsets(hulk_ctx *c, const uint64_t *h_offsets);

}  // namespace hulk
