/*
 * hulk_hip.h — C ABI of libhulkhip.so: the MI355X (gfx950) implementation of HULK's
 * `sketch` hot path (minimizers -> jump-hash k-mer spectrum -> count-min + CWS histosketch).
 *
 * The reference (will-rowe/hulk v1.0.0) has no FFI layer; the seam this library replaces is
 * the Go package API that `SeqMinimizer.Run` / `Sketcher.Run` drive:
 *
 *   reference interface (file:line)                               this ABI
 *   -----------------------------------------------------------   -------------------------
 *   pipeline.findMinimizers(chan, *Info) (*theBoss, error)        hulk_create
 *       src/pipeline/boss.go:54
 *   histosketch.NewHistoSketch(k, s, bins, decay)                 hulk_create (same checks,
 *       src/histosketch/histosketch.go:50-92                        same error strings)
 *   theBoss.AddSeq(seq []byte)        src/pipeline/boss.go:24-26  hulk_add_reads[_device]
 *   theBoss.Flush()                   src/pipeline/boss.go:34-36  hulk_flush
 *   theBoss.StopWork() + final flush  src/pipeline/boss.go:29-31, hulk_finish
 *       src/pipeline/sketch.go:219-224
 *   theBoss.GetMinimizerCount()       src/pipeline/boss.go:39-41  hulk_get_counters
 *   HistoSketch.AddElement(bin, v)    src/histosketch/histosketch.go:129-155
 *                                       (driven internally by hulk_flush; exposed for
 *                                        tests as hulk_add_histogram)
 *   HistoSketch.Sketch / .SketchWeights (exported fields)         hulk_get_sketch
 *       src/histosketch/histosketch.go:40-41
 *   KmerSpectrum bins                 src/kmerspectrum/kmerspectrum.go:25
 *                                                                 hulk_get_histogram
 *   interval rule `seqCount % Interval == 0`                      params.interval
 *       src/pipeline/sketch.go:211-215
 *   DataStreamer.Run + FastqHandler.Run (lines -> reads)          hulk_parse_files (host only),
 *       src/pipeline/sketch.go:40-79, 99-161; seqio.go:38-40        hulk_sketch_files (+ the AddSeq
 *                                                                   loop of sketch.go:196-217)
 *   SeqMinimizer.Run's AddSeq / Flush loop when the read stream   hulk_comm_init + hulk_step_sharded
 *       is sharded over several GPUs                                (+ hulk_gather_sketch for
 *       src/pipeline/sketch.go:182-250, boss.go:24-41               Sketcher.Run's result, sketch.go:271-301)
 *
 * Conventions: every call returns HULK_OK (0) or a negative HULK_ERR_*; the message the
 * reference would have passed to log.Fatalf("ERROR---> %v") is available from
 * hulk_last_error()/hulk_strerror().  The library never aborts the process.  A context is
 * single-caller (the reference's SeqMinimizer.Run is one goroutine).  Caller owns every
 * buffer it passes; host buffers may be released as soon as the call returns.  Work is
 * asynchronous on one HIP stream; hulk_finish / hulk_synchronize / hulk_get_* are the synchronisation points and
 * the place where device-side errors (short read, <1% bins used) surface.
 */
#ifndef HULK_HIP_H
#define HULK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: hulk_params.reserved[0] became `flags` (unknown bits are refused), hulk_set_profiling takes a mask, the multi-GPU
 * entry points (hulk_comm_*, hulk_step_*, hulk_gather_sketch).
 * 3: every knob a host may want per context is a field (as SketchCmd carries every flag as a field,
 *    src/pipeline/pipeline.go:14-30): hulk_params.batch / work_lanes / host_copy_threads (were reserved[0..2], 0 = default),
 *    HULK_FLAG_SHARD_FULL / HULK_FLAG_NO_OVERLAP, hulk_ingest_opts with hulk_parse_files_opts / hulk_sketch_files_opts.
 *    The HULK_* environment variables that remain are overrides for profiling scripts, read when a context is created.
 * 4: HULK_FLAG_CMS_CHAIN + hulk_get_device_checks (the count-min replay's hardware assumption is verified on the device by
 *    hulk_create), hulk_get_profile_table (hulk_set_profiling bit 32), hulk_load_sketches / hulk_smash_files (the directory form
 *    of `hulk smash`); the test hooks hulk_debug_inject / hulk_debug_read left the shipping library (profiling build only).
 * Bindings compare it with the value they were written for. */
#define HULK_ABI_VERSION 4

#define HULK_OK 0
#define HULK_ERR_W (-1)          /* "w must be: 0 < w < 257"                  minimizer.go:63 */
#define HULK_ERR_K (-2)          /* "k size must be: 0 < k < 32"              minimizer.go:66 */
#define HULK_ERR_EMPTY_SEQ (-3)  /* "sequence length must be > 0"             minimizer.go:72 */
#define HULK_ERR_SHORT_SEQ (-4)  /* "sequence length must be >= w + k - 1"    minimizer.go:75 */
#define HULK_ERR_FEW_BINS (-5)   /* "not used yet"                            kmerspectrum.go:95 */
#define HULK_ERR_HS_K (-6)       /* "histosketching only supports k <= 31"    histosketch.go:54 */
#define HULK_ERR_DECAY (-7)      /* "decay ratio must be between 0.0 and 1.0" histosketch.go:63 */
#define HULK_ERR_BINS (-8)       /* "histogram must have at least 2 bins"     histosketch.go:66 */
#define HULK_ERR_NEG_BINS (-9)   /* "negative value used for number of k-mer spectrum bins: %d" kmerspectrum.go:34 */
#define HULK_ERR_NO_SEQ (-10)    /* "no sequences received"                   pipeline/sketch.go:238 */
#define HULK_ERR_FASTQ_ID (-11)  /* "read ID in fastq file does not begin with @: %v"  seqio.go:39 */
#define HULK_ERR_LINE_TOO_LONG (-12) /* "bufio.Scanner: token too long" (a line of >= 64 KiB; sketch.go:53,76 log.Fatal(scanner.Err())) */
#define HULK_ERR_ARG (-30)       /* bad argument to this ABI (NULL pointer, bad shard, ...) */
#define HULK_ERR_HIP (-31)       /* HIP runtime failure; hulk_last_error has the hipError string */
#define HULK_ERR_NO_DEVICE (-32) /* no usable gfx950 device */
#define HULK_ERR_READ_TOO_LONG (-33) /* read longer than this build's per-read limit */
#define HULK_ERR_STATE (-34)     /* call not valid in this state (e.g. add after finish) */
#define HULK_ERR_IO (-35)        /* open/read/gzip failure; the message is the one Go's os/gzip error carries */
#define HULK_ERR_FASTA_HEADER (-36) /* --fasta input without any '>' line (the reference panics on l1[0] = 64, sketch.go:127) */
#define HULK_ERR_COMM (-37)      /* RCCL (or the host's exchange function) failed; hulk_last_error has the detail */

/* How the CWS parameter matrices r, c, b (histosketch.go:95-126) are produced. */
#define HULK_CWS_GO_COMPAT 0     /* go_rng Gamma/Uniform over Go math/rand, seed 1 (default) */
#define HULK_CWS_EXTERNAL 1      /* caller supplies them with hulk_set_cws_tables (e.g. dumped by a Go program) */

/* hulk_params.flags.  None of the NO_* switches changes a result; they only remove an exact shortcut so that
 * its effect can be measured (bench.py --no-prune) and tested (tests/test_gpu_parity.py). */
#define HULK_FLAG_GAMMA_CPYTHON 1u /* go_rng's Gamma (third-party, github.com/leesper/go_rng @ a612b043e353, not vendored by
                                    * the reference: histosketch.go:103,112-113) is restated from memory as CPython's
                                    * gammavariate with the squeeze constant 4*exp(-0.5)/sqrt(2) (the default here); this flag
                                    * selects CPython's own SG_MAGICCONST = 1 + ln 4.5.  Any valid squeeze only short-cuts the
                                    * real test r >= ln z, so the tables are the same either way (tested); the flag exists so
                                    * that a pin against a real Go run can be reproduced literally */
#define HULK_FLAG_NO_PRUNE 2u      /* CWS scan reads the whole table for every interval (no per-tile bound, no whole-batch bound) */
#define HULK_FLAG_NO_SKIP 4u       /* keep the per-tile bound, drop the whole-batch bound */
#define HULK_FLAG_SHARD_FULL 8u    /* hulk_step_sharded always exchanges the k-mer spectra (never the count-min increments) */
#define HULK_FLAG_NO_OVERLAP 16u   /* one stream: flush kernels on the work stream, one work lane (profiling: every kernel
                                    * runs alone, so its own duration can be read) */
#define HULK_FLAG_CMS_CHAIN 64u    /* the count-min replay (countmin.go:103-147 in bin order) runs its chain-form kernels: the order of
                                    * the bins inside a 64-bin chunk comes from a static table and register exchanges instead of from
                                    * the order in which the LDS applies the lanes of one returning atomic add.  hulk_create selects
                                    * them by itself on a device whose LDS does not keep that order (hulk_get_device_checks); same
                                    * additions in the same order: bit-identical counters, estimates and sketch (tested) */
#define HULK_FLAG_NO_PRERESERVE 32u /* hulk_create does not size the work lanes' minimizer lists for batch x interval reads (about
                                     * 3 GB per lane at the defaults): they grow with the first batches instead, at the price of an
                                     * allocation in the middle of the stream.  For contexts that see few or long reads, or many
                                     * contexts (ranks) on one GPU. */

/* Largest k-mer spectrum this build bins: the binning kernels pack (spectrum slot << 20 | bin) into one dword
 * (k^4 = 923,521 < 2^20 at the reference's maximum k = 31; cmd/sketch.go:118). */
#define HULK_MAX_BINS (1 << 20)

typedef struct hulk_ctx hulk_ctx;

typedef struct hulk_params {
    uint32_t k;            /* -k/--kmerSize   (cmd/root.go:62)   */
    uint32_t w;            /* -w/--windowSize (cmd/sketch.go:52) */
    uint32_t sketch_size;  /* -s/--sketchSize (cmd/sketch.go:54) */
    int32_t  num_bins;     /* 0 => Pow(k,4) as cmd/sketch.go:118 */
    double   decay_ratio;  /* -x/--decayRatio (cmd/sketch.go:55); 1.0 = concept drift off */
    uint32_t interval;     /* -i/--interval   (cmd/sketch.go:53); 0 = none */
    int32_t  device;       /* HIP device ordinal */
    uint32_t slot_begin;   /* this context owns sketch slots [slot_begin, slot_begin+slot_count) */
    uint32_t slot_count;   /* 0 => all slots (single-GPU) */
    uint32_t cws_source;   /* HULK_CWS_* */
    uint32_t flags;        /* HULK_FLAG_* (0 = defaults) */
    uint32_t batch;        /* sketching intervals binned by one launch chain and flushed by ONE pass over the CWS table:
                            * 1..16, 0 = default (16).  Any value gives the same sketch (hulk_batch_size) */
    uint32_t work_lanes;   /* 2: consecutive batches are binned on two alternating work streams, so that the minimizer kernel
                            * of a batch runs beside the jump-hash / spectrum kernels of the batch before it; 1: one work
                            * stream; 0 = default: 2, but 1 with count-min decay (0 < decay_ratio < 1), whose flush needs the
                            * CUs a second lane would keep occupied.  A context on a caller's stream (hulk_set_stream) joins the second
                            * lane into that stream at the end of every call — the caller still sees ONE stream — and so
                            * gives up that overlap */
    uint32_t host_copy_threads; /* threads that copy a chunk of hulk_add_reads' host buffers into pinned staging: 1..32,
                            * 0 = default (4) */
    uint32_t reserved;     /* must be 0 */
} hulk_params;

/* Version of this ABI (HULK_ABI_VERSION). */
int hulk_abi_version(void);
/* "abi=4 arch=gfx950 sources=<first 16 hex digits of the SHA-256 over hulk_amd/csrc's sources and headers, in the Makefile's
 * order> hipcc=<version>[ experiments=1]": which tree and compiler this .so was built from (a binding or a test can refuse a
 * stale one); " experiments=1" marks the profiling build (make EXPERIMENTS=1: experiment switches and test hooks compiled in). */
const char *hulk_build_info(void);
/* Reference error text for a status code. */
const char *hulk_strerror(int status);
/* Message of the last failure on this context ("" if none). NULL ctx => last hulk_create failure. */
const char *hulk_last_error(const hulk_ctx *ctx);

/* findMinimizers + NewHistoSketch.  Allocates device state, generates/uploads the CWS tables.  With an interval, the work
 * lanes' buffers are sized here for the largest batch the interval rule forms (batch x interval reads, ~3 GB per lane at
 * the defaults): no call of the stream allocates device memory.  (interval = 0: the first call sizes them.) */
int hulk_create(const hulk_params *params, hulk_ctx **out);
void hulk_destroy(hulk_ctx *ctx);

/* Run all work on the caller's hipStream_t (e.g. torch's current stream) instead of the
 * context's own stream.  NULL is the HIP null (default) stream. */
int hulk_set_stream(hulk_ctx *ctx, void *hip_stream);
/* Go back to the context's private non-blocking stream (the default after hulk_create). */
int hulk_set_private_stream(hulk_ctx *ctx);

/* Supply r, c, b ([sketch_size][num_bins] row-major fp64, full matrices, host memory) when
 * cws_source == HULK_CWS_EXTERNAL.  Must be called before the first read. */
int hulk_set_cws_tables(hulk_ctx *ctx, const double *r, const double *c, const double *b);

/* AddSeq for a batch.  Read i is bases[offsets[i] .. offsets[i+1]) (ASCII, any case).
 * The interval rule is applied inside: a flush is queued whenever the global read count hits a
 * multiple of params.interval.  The host variant validates the lengths and has copied the caller's
 * buffers (into pinned staging, in chunks) when it returns — they may be reused at once (the cgo rule);
 * the PCIe copies and the kernels are queued on the context's stream and may still be running:
 * hulk_flush / hulk_finish / the getters are the synchronisation points. */
int hulk_add_reads(hulk_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads);
/* Same, buffers already resident in this device's memory (offsets too).  `max_read_len` is an
 * upper bound on the read lengths in the batch (selects the kernel configuration);
 * `bases_bytes` is the size of the bases allocation.  Length validation happens on device.
 * Stream order: the kernels run on the context's stream (private and non-blocking unless
 * hulk_set_stream was called), so the buffers must be complete before the call — produce them on
 * that stream or synchronise first — and must stay alive until the stream has passed the call. */
int hulk_add_reads_device(hulk_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_offsets,
                          uint64_t n_reads, uint32_t max_read_len, uint64_t bases_bytes);

/* ---- host ingest: DataStreamer.Run + FastqHandler.Run ------------------------
 * Inputs are read in order (n_paths == 0: STDIN; a name whose last '.'-separated element is "gz" is
 * gunzipped), cut into lines the way bufio.Scanner/ScanLines does ('\n', one trailing '\r' dropped,
 * unterminated last line kept, a line of >= 65536 bytes is HULK_ERR_LINE_TOO_LONG) and grouped into
 * reads by the reference's slot machine: FASTQ = l1..l3 take the next NON-EMPTY line, l4 takes the
 * next line whatever it is, the read is l2 and its l1 must start with '@' (checked when l4 arrives,
 * so a truncated last record is dropped silently); --fasta = lines of a '>' record concatenated,
 * parsing stops at the first empty line.  `threads` parser threads (0 = one per core, at most 16; see hulk_ingest_opts).
 * gzip input is inflated by the library itself, multistream as compress/gzip reads it: a regular one-member file of >= 4 MiB
 * by hulk_ingest_opts.gz_threads (default 16) threads at once (about 0.6 GB of scratch mappings while the file is open;
 * HULK_INGEST_GZ_ONE_THREAD: one thread), a bgzip'd file member by member on as many; bytes and messages are the same whichever reader runs. */
typedef struct hulk_ingest_stats {
    uint64_t n_seqs;      /* seqCount of SeqMinimizer.Run */
    uint64_t total_len;   /* lengthTotal */
    uint64_t n_lines;
    uint64_t bytes_in;    /* bytes read (after gunzip) */
    double   seconds;
} hulk_ingest_stats;
/* Called once per parsed batch, in input order; buffers are only valid during the call. */
typedef int (*hulk_batch_fn)(void *user, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads);
/* Parse only (no GPU, no context): every batch goes to `fn`.  Error text -> errbuf. */
int hulk_parse_files(const char *const *paths, uint32_t n_paths, int fasta, uint32_t threads,
                     hulk_batch_fn fn, void *user, hulk_ingest_stats *stats, char *errbuf, uint64_t errbuf_len);
/* The knobs of a run of the host ingest, as fields (0 = the default everywhere).  hulk_parse_files / hulk_sketch_files take
 * the defaults (and `threads` as parser_threads); the HULK_INGEST_* / HULK_GZ_* environment variables, where set, override
 * either — they exist for profiling scripts and tests. */
#define HULK_INGEST_GZ_ONE_THREAD 1u  /* every gzip input through the one-thread reader (no parallel member / BGZF readers) */
#define HULK_INGEST_GZ_ZLIB 2u        /* zlib's inflate instead of the library's own decoder */
#define HULK_INGEST_TRACE 4u          /* seconds per phase of the calling thread and of the gzip readers, on stderr */
#define HULK_INGEST_HOST_PARSER 8u    /* hulk_sketch_files*: lines -> sequences on the host's parser threads.  Default: the raw file
                                       * bytes go to the GPU as they are read and the line machine runs there (hulk_fastq.hip).  FASTQ: a
                                       * block the device will not decide — a header without '@', a line of 64 KiB, an over-long run of
                                       * empty lines — hands the stream to the host parser, whose reads and messages are the same.
                                       * --fasta: the device decides everything (the empty line that ends the parsing, the line of 64 KiB
                                       * that is bufio.Scanner's error); sequences of any length accumulate in device memory across blocks */
typedef struct hulk_ingest_opts {
    uint32_t parser_threads;  /* 0 = one per hardware thread, at most 16 (the measured optimum); any other figure is taken as it is (<= 256) */
    uint32_t gz_threads;      /* threads inflating the members of a bgzip'd input, or the chunks of ONE gzip member, side by side: 1..64, 0 = 16 */
    uint32_t file_readers;    /* pieces a block of a regular file is read in, side by side: 1..16, 0 = 4 */
    uint32_t flags;           /* HULK_INGEST_* */
    uint64_t block_bytes;     /* bytes per block of the line pump: >= 128 KiB, 0 = 32 MiB */
    uint64_t gz_chunk_bytes;  /* compressed bytes per chunk of the parallel one-member reader: >= 8 KiB, 0 = 1 MiB */
    uint64_t reserved[2];     /* must be 0 */
} hulk_ingest_opts;
int hulk_parse_files_opts(const char *const *paths, uint32_t n_paths, int fasta, const hulk_ingest_opts *opts,
                          hulk_batch_fn fn, void *user, hulk_ingest_stats *stats, char *errbuf, uint64_t errbuf_len);
/* Parse and AddSeq every read (pinned double-buffered staging, copies and kernels asynchronous on
 * the context's stream while the next block is read and parsed).  The interval rule applies as in
 * hulk_add_reads; the caller still ends the run with hulk_finish (Flush + StopWork). */
int hulk_sketch_files(hulk_ctx *ctx, const char *const *paths, uint32_t n_paths, int fasta, uint32_t threads,
                      hulk_ingest_stats *stats);
int hulk_sketch_files_opts(hulk_ctx *ctx, const char *const *paths, uint32_t n_paths, int fasta, const hulk_ingest_opts *opts,
                           hulk_ingest_stats *stats);

/* Intervals are flushed in batches: up to hulk_batch_size() consecutive sketching intervals are
 * binned into separate k-mer spectra by one kernel launch and then pushed through count-min + CWS
 * with ONE pass over the CWS table.  The result is identical to flushing them one at a time. */
uint32_t hulk_batch_size(const hulk_ctx *ctx);

/* Multi-GPU split: (1) bin this rank's reads WITHOUT applying the interval rule: read i goes to
 * spectrum i / reads_per_spectrum (0 => all into spectrum 0; at most hulk_batch_size() spectra),
 * (2) the caller all-reduces hulk_histogram_device() across ranks (RCCL, uint32 sum,
 * n_spectra * num_bins contiguous elements), (3) hulk_flush_batch(n_spectra) on every rank. */
int hulk_bin_reads_device(hulk_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_offsets,
                          uint64_t n_reads, uint32_t max_read_len, uint64_t bases_bytes,
                          uint64_t reads_per_spectrum);
/* The same with the first read going to spectrum `first_spectrum` instead of 0 (needs reads_per_spectrum > 0): a rank
 * that bins WHOLE intervals of a batch — intervals [first_spectrum, first_spectrum + n_reads / reads_per_spectrum) —
 * while the other ranks bin the others; the all-reduce over the ring then is a gather (the other spectra are zero here). */
int hulk_bin_reads_device_at(hulk_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_offsets,
                             uint64_t n_reads, uint32_t max_read_len, uint64_t bases_bytes,
                             uint64_t reads_per_spectrum, uint32_t first_spectrum);
uint32_t *hulk_histogram_device(hulk_ctx *ctx);
int hulk_flush_batch(hulk_ctx *ctx, uint32_t n_spectra);
/* Same, but the flush waits for the work queued so far on `dep_stream` (the stream the all-reduce
 * was issued on) instead of the context's work stream, so the next hulk_bin_reads_device on the
 * work stream runs UNDER the collective.  The caller must have made dep_stream wait for the binning
 * (event / wait_stream) before the collective. */
int hulk_flush_batch_after(hulk_ctx *ctx, uint32_t n_spectra, void *dep_stream);

/* ---- multi-GPU with the exchange INSIDE the library (one context = one rank = one GPU) -------------------------
 * The seam is still SeqMinimizer.Run (src/pipeline/sketch.go:182-250) driving theBoss (boss.go:24-41): a host that
 * shards the read stream over G processes calls hulk_comm_init once and then hulk_step_sharded instead of
 * hulk_add_reads_device; the interval rule stays the reference's (a flush every `interval` reads of the GLOBAL stream,
 * sketch.go:211-215) and the sketch is the one a single GPU computes over the same stream.
 *
 * Step s of a G-rank run covers the global sketching intervals [s*G*T, (s+1)*G*T), T = hulk_batch_size(); rank g owns
 * the WHOLE intervals [s*G*T + g*T, s*G*T + (g+1)*T) — one contiguous chunk of T*interval reads — and the sketch
 * slots [slot_begin, slot_begin + slot_count) given at hulk_create (count-min is replicated, the CWS update is
 * slot-sharded).  Per step the library picks one of two exact exchanges on its flush stream:
 *   full   (the first step of a stream, concept drift / decay, or while any rank's whole-step bound says an element
 *           could still lower one of its slots' weights): all-gather of the G*T k-mer spectra, then the ordinary flush
 *           of every rank's T intervals in stream order on every rank;
 *   delta  (steady state: count-min counters only grow and AddElement only replaces a weight by a smaller one,
 *           countmin.go:103-138, histosketch.go:139-153, so once every rank's bound min_row(K)/min(counter) cannot get
 *           below any of its weights, no element can change the sketch): each rank reduces ITS intervals to the
 *           count-min increments they cause (7 x 2000 integers per interval) and their used-bin counts (the 1 % rule,
 *           kmerspectrum.go:84-96), ONE all-gather of those (G*T*56 KB instead of G*T*k^4*4 B), every rank adds them in
 *           stream order.  The bound is evaluated at the start of step s on every rank and travels with step s's
 *           exchange; it governs step s+1 (counters and weights are monotone, so a bound that held one step earlier
 *           still holds).  Integer sums: bit-identical to the full exchange (tested on both paths).
 * The last step of a stream may be ragged: `step_intervals` (the same value on every rank) is the number of intervals
 * of the global stream in this step, rank g holds min(T, max(0, step_intervals - g*T)) of them and the last one may
 * be partial (the reference's EOF flush, sketch.go:219-221).  hulk_finish afterwards only synchronises (and reports
 * "no sequences received", sketch.go:237-239, only if the GLOBAL stream was empty: a rank may hold none of a short stream).
 * Failure: HULK_ERR_ARG / HULK_ERR_STATE are raised before anything is queued and leave the context as it was.  HULK_ERR_HIP /
 * HULK_ERR_COMM from a step are FATAL for the communicator and the run — the peers may be inside a collective this rank has
 * left — and every later call on the context returns the same status; the host ends all ranks. */
#define HULK_UNIQUE_ID_BYTES 128
/* ncclGetUniqueId: called by ONE rank; the host hands the 128 bytes to the other ranks over any channel it has. */
int hulk_comm_unique_id(void *unique_id);
/* ncclCommInitRank (RCCL over xGMI; librccl.so.1 is bound when this is first called).  Collective call. */
int hulk_comm_init(hulk_ctx *ctx, const void *unique_id, uint32_t rank, uint32_t world);
/* The same protocol over a transport the HOST provides (hosts without RCCL between their ranks; the test suite runs two
 * ranks on one GPU this way, which RCCL refuses): the library stages the buffers through pinned host memory and calls
 * `fn` synchronously.  op HULK_XCHG_ALLGATHER: send = this rank's `bytes`, recv = world * bytes, rank order;
 * op HULK_XCHG_ALLREDUCE_U32: element-wise uint32 sum of `bytes` / 4 words over the ranks, send -> recv. */
#define HULK_XCHG_ALLGATHER 0
#define HULK_XCHG_ALLREDUCE_U32 1
typedef int (*hulk_exchange_fn)(void *user, int op, const void *send, void *recv, uint64_t bytes);
int hulk_comm_init_host(hulk_ctx *ctx, uint32_t rank, uint32_t world, hulk_exchange_fn fn, void *user);
/* Projection aid (tools/shard_projection.py): no peer at all — the other ranks' contributions to an all-gather are
 * copies of this rank's own, an all-reduce is the identity.  One GPU then runs exactly one rank's share of a G-rank
 * step, which bounds the G-GPU rate from above. */
int hulk_comm_init_loopback(hulk_ctx *ctx, uint32_t rank, uint32_t world);
/* One step of this rank: bin its whole intervals of the step (n_reads <= T * interval reads, resident in HBM as for
 * hulk_add_reads_device), exchange, flush.  Asynchronous; the next step's binning runs under this step's exchange. */
int hulk_step_sharded(hulk_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                      uint32_t max_read_len, uint64_t bases_bytes, uint32_t step_intervals);
/* The same for reads in HOST memory (a Go host's slices): validated as hulk_add_reads does, copied into pinned staging
 * before the call returns (the cgo rule), PCIe copy and kernels queued on the context's stream. */
int hulk_step_sharded_host(hulk_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                           uint32_t step_intervals);
/* SURVEY.md 8(e) to the letter, for comparison: every rank bins `reads_per_spectrum` reads of each of `n_spectra`
 * intervals (its slice of every interval), ONE all-reduce (uint32 sum) of the n_spectra spectra, flush. */
int hulk_step_sliced(hulk_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                     uint32_t max_read_len, uint64_t bases_bytes, uint64_t reads_per_spectrum, uint32_t n_spectra);
/* Sketcher.Run's result on every rank (src/pipeline/sketch.go:271-301): all-gather of the ranks' slot shards. */
int hulk_gather_sketch(hulk_ctx *ctx, uint64_t *mins, double *weights);
/* Steps taken by each exchange and the bytes this rank received through the transport (cumulative). */
int hulk_get_comm_stats(hulk_ctx *ctx, uint64_t *steps_delta, uint64_t *steps_full, uint64_t *bytes_received);
/* Health of the exchange headers (cumulative): `refetched` = times this rank's view of a header (its own block in the host
 * transport's staging, or the gathered header of the previous step) was not there when its copy / event said so and was
 * taken again after a synchronisation; `void_blocks` = times a rank's block of the previous step carried another step's seal
 * (every rank then takes the spectra exchange; after a delta step the run ends with HULK_ERR_COMM).  Both are 0 in a healthy run. */
int hulk_get_comm_health(hulk_ctx *ctx, uint64_t *refetched, uint64_t *void_blocks);
/* ---- test hooks: compiled into the PROFILING build only (make -C hulk_amd/csrc EXPERIMENTS=1 -> libhulkhip_exp.so, which
 * defines HULK_EXPERIMENTS; hulk_build_info() then ends in " experiments=1").  The shipping libhulkhip.so does not export them. */
#ifdef HULK_EXPERIMENTS
/* Test hook (tests/test_gpu_two_rank.py): at step `step` of hulk_step_sharded this rank
 *   HULK_INJECT_STALE_SEAL   seals its header block with the PREVIOUS step's tag (a block that is not of this step);
 *   HULK_INJECT_STALE_STAGE  (host transport) finds its own block missing from the host staging on the first attempt. */
#define HULK_INJECT_NONE 0u
#define HULK_INJECT_STALE_SEAL 1u
#define HULK_INJECT_STALE_STAGE 2u
int hulk_debug_inject(hulk_ctx *ctx, uint32_t what, uint64_t step);
/* Test hook: internal buffers of the CWS scan as the latest flush left them (after a synchronisation).
 *   HULK_DEBUG_TILEMIN  float[slot groups][wave tiles][8]: the fp32 tile minima of the batch (no concept drift: plane 0)
 *   HULK_DEBUG_SCANMAP  uint64[slot groups][(wave tiles + 63) / 64]: which tiles the scan read
 * *bytes_io: capacity of `out` in, bytes written out (HULK_ERR_ARG if too small; the size needed is returned in it). */
#define HULK_DEBUG_TILEMIN 1u
#define HULK_DEBUG_SCANMAP 2u
int hulk_debug_read(hulk_ctx *ctx, uint32_t what, void *out, uint64_t *bytes_io);
#endif /* HULK_EXPERIMENTS */

/* Process-level buffers the library keeps between calls — the device FASTQ parser's pinned and device blocks (hulk_sketch_files:
 * 96 MB pinned and about 300 MB of HBM per set at the default block size; at most two sets, and a set nobody borrowed for 10 s is
 * freed by the process's next hulk_create / hulk_destroy / hulk_sketch_files; a set that has parsed --fasta input also holds a
 * two line indices for 2-byte lines and two accumulation buffers that grow with the longest sequence: about 0.8 GB of HBM more), the
 * host parsers' large buffers (gzip, FASTA, the
 * host FASTQ parser: ordinary memory, at most 1 GiB, same 10 s rule) and hulk_smash's device arrays — are freed at once;
 * the next call that needs them allocates again.  Call it with no hulk_sketch_files / hulk_smash in flight on another thread. */
int hulk_release_caches(void);

/* Test hook: add counts to the current k-mer spectrum directly (host uint32[num_bins]). */
int hulk_add_histogram(hulk_ctx *ctx, const uint32_t *bins);

/* theBoss.Flush(): histosketch the current k-mer spectrum, then wipe it. */
int hulk_flush(hulk_ctx *ctx);
/* Final flush + StopWork; synchronises and reports any deferred device-side error. */
int hulk_finish(hulk_ctx *ctx);

/* Wait until everything queued so far — copies, binning kernels, flushes — has run (the getters below do the same). */
int hulk_synchronize(hulk_ctx *ctx);

/* Outputs (each synchronises the stream).  mins/weights are full length sketch_size; slots not
 * owned by this context keep their initial values (0 / MaxFloat64). */
int hulk_get_sketch(hulk_ctx *ctx, uint64_t *mins, double *weights);
int hulk_get_counters(hulk_ctx *ctx, uint64_t *n_reads, uint64_t *n_minimizers, uint64_t *total_len);
int hulk_get_histogram(hulk_ctx *ctx, uint32_t *bins);
/* Count-min counters as fp64 [7][2000] (test hook). */
int hulk_get_cms(hulk_ctx *ctx, double *counters);
/* Copy rows of the CWS tables owned by this context to host: each [slot_count][num_bins] (test hook). */
int hulk_get_cws_tables(hulk_ctx *ctx, double *r, double *c, double *b);

/* `hulk smash` (cmd/smash.go:183-226): pairwise distance matrix of n_sketches histosketches of
 * sketch_size slots (host arrays, row-major).  distances[s * n + q] = HULKdata.GetDistance(subject s,
 * query q) (src/sketchio/sketchio.go:259-306): jaccard = distances.go:19-26, weighted jaccard =
 * distances.GetWJD (distances.go:44-72) with the subject's weights on both sides, as the reference does. */
#define HULK_METRIC_JACCARD 0
#define HULK_METRIC_WEIGHTED_JACCARD 1
int hulk_smash(int device, const uint64_t *mins, const double *weights, uint32_t n_sketches,
               uint32_t sketch_size, int metric, double *distances);
/* The same; *kernel_ms (may be NULL) receives the duration of the distance kernel alone (HIP events), without the PCIe copies. */
int hulk_smash_ex(int device, const uint64_t *mins, const double *weights, uint32_t n_sketches,
                  uint32_t sketch_size, int metric, double *distances, double *kernel_ms);

/* What hulk_create verified on this context's device.  *lds_order_ok: 1 if one returning LDS atomic add (ds_add_rtn_u64 /
 * ds_add_rtn_f64) applies the lanes that hit one address in ascending lane order and a wave's instructions in program order —
 * the property the count-min replay kernels take the reference's bin order from (countmin.go:103-138); checked once per
 * process and device with all-equal, paired and pseudo-random address patterns.  *cms_chain_form: 1 if this context runs the
 * chain-form replay kernels (the property does not hold, or HULK_FLAG_CMS_CHAIN). */
int hulk_get_device_checks(hulk_ctx *ctx, uint32_t *lds_order_ok, uint32_t *cms_chain_form);

/* ---- `hulk smash`, the directory form (cmd/smash.go:160-226), native: LoadHULKdata for every file (sketchio.go:100-195: JSON,
 * class / version, the MD5 of the little-endian mins against the stored md5sum, helpers.go:156-166) on `threads` host threads
 * (0 = one per hardware thread, at most 32), FindSketch(ksize, algo) per file (sketchio.go:198-257), the equal-length check of
 * GetDistance (sketchio.go:274-277).  Paths are taken in sort.Strings order (a path given twice counts once: the reference keeps
 * them in a map).  Failures: HULK_ERR_ARG with the reference's text in `errbuf` (and hulk_last_error(NULL)) — of the first file
 * in sorted order that fails to load; then "<n> sketches found ... needs at least 2"; then of the first file whose FindSketch
 * fails; then the length mismatch.  hulk_load_sketches needs no GPU. */
typedef struct hulk_sketch_set hulk_sketch_set;
int hulk_load_sketches(const char *const *paths, uint32_t n_paths, uint32_t ksize, const char *algo, uint32_t threads,
                       hulk_sketch_set **out, char *errbuf, uint64_t errbuf_len);
void hulk_sketch_set_free(hulk_sketch_set *set);
int hulk_sketch_set_info(const hulk_sketch_set *set, uint32_t *n_sketches, uint32_t *sketch_size);
const uint64_t *hulk_sketch_set_mins(const hulk_sketch_set *set);      /* [n_sketches][sketch_size], sorted-path order */
const double *hulk_sketch_set_weights(const hulk_sketch_set *set);     /* the same shape (zeros for kmv / khf: they carry no weights) */
const char *hulk_sketch_set_path(const hulk_sketch_set *set, uint32_t i);
const char *hulk_sketch_set_banner(const hulk_sketch_set *set, uint32_t i);   /* banner_label of file i */
typedef struct hulk_smash_stats {
    double seconds_load;     /* reading, parsing and verifying the files */
    double seconds_matrix;   /* hulk_smash (copies + kernels) */
    double seconds_csv;      /* formatting and writing the CSV file(s) */
    double kernel_ms;        /* the distance kernel alone */
    uint32_t n_sketches, sketch_size;
} hulk_smash_stats;
/* runSmash + makeMatrix: load, smash on `device`, and — matrix_csv_path != NULL — write the file encoding/csv would: the sorted
 * paths as header, then one row per subject of strconv.FormatFloat(100 - 100 * distance, 'f', 2, 64) (cmd/smash.go:183-226);
 * banner_csv_path != NULL: makeBannerMatrix's file (cmd/smash.go:229-261: a sketch's mins + its banner label per line, in
 * sorted file order).  metric: "jaccard" | "weightedjaccard"; algo: "histosketch" | "kmv" | "khf".  `distances` (may be NULL)
 * receives [n][n] in sorted-path order; `stats` may be NULL. */
int hulk_smash_files(int device, const char *const *paths, uint32_t n_paths, uint32_t ksize, const char *algo, const char *metric,
                     uint32_t threads, const char *matrix_csv_path, const char *banner_csv_path, double *distances,
                     hulk_smash_stats *stats, char *errbuf, uint64_t errbuf_len);

/* Device self-test: the jump hash replaces the fp64 division 2^31/r by a Newton reciprocal; this
 * checks RN(1/r) against IEEE division for EVERY r in [1, 2^31] and returns the mismatch count. */
int hulk_selftest_reciprocal(hulk_ctx *ctx, uint64_t *mismatches);

/* Exact pruning of the CWS table scan (no concept drift): a 256-bin x 8-slot tile of K is only read when
 * min(K) * max(1/f) (or min(K) * min(1/f) for min(K) >= 0) can get below one of its slots' current weights —
 * AddElement only replaces a weight by a smaller A (histosketch.go:139-153), so the sketch is unchanged.
 * Reports how many tiles the scans read out of how many they covered (both cumulative). */
int hulk_get_scan_stats(hulk_ctx *ctx, uint64_t *tiles_visited, uint64_t *tiles_total);

/* Per-kernel timing for bench.py: when enabled, hipEvents bracket every launch of the heavy kernels
 * ("k_minimizer_fast", "k_jump_bin", "k_jump_left", "k_cws_scan", "k_cmsd_freq"; each alone) on the stream they are launched on.
 * enabled: 0 off, 1 all of them, otherwise a mask (2 k_minimizer_fast, 4 k_jump_bin and k_jump_left, 8 k_cws_scan, 16 k_cmsd_freq,
 * 32 every launch: hulk_get_profile_table)
 * — every bracketed launch costs the stream two event records (~3 % of a C2 step for all of them), so the timed pass of
 * bench.py brackets nothing and the durations come from a separate pass. */
int hulk_set_profiling(hulk_ctx *ctx, int enabled);
/* Number of timed launches of `kernel` and their summed duration (synchronises; clears that log). */
int hulk_get_profile(hulk_ctx *ctx, const char *kernel, uint64_t *launches, double *total_ms);
/* hulk_set_profiling bit 32: EVERY kernel launch of the step path (binning chain and flush) is timed — an event in front of each
 * launch, a kernel's duration = the time to the next event on its stream — meant for HULK_FLAG_NO_OVERLAP contexts, where every
 * kernel runs alone.  Writes "kernel<TAB>launches<TAB>total_ms<LF>" lines (NUL-terminated) into `out` (HULK_ERR_ARG if `cap`
 * is too small), synchronises, clears the log. */
int hulk_get_profile_table(hulk_ctx *ctx, char *out, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* HULK_HIP_H */
