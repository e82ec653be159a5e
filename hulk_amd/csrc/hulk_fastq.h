// hulk_fastq.h — the device FASTQ parser's interface between its kernels (hulk_fastq.hip) and the ingest (hulk_ingest.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hulk {

constexpr uint32_t FQ_MAX_TOKEN = 64 * 1024;   // bufio.MaxScanTokenSize: a line of this many bytes ends the scan
// reasons a block is handed to the host parser (FqState.need_host)
constexpr uint32_t FQ_NEED_TAIL = 1;    // the bytes behind the last completed record of the previous block outgrow the porch
constexpr uint32_t FQ_NEED_LINES = 2;   // more lines / reads / bases than the index arrays hold
constexpr uint32_t FQ_NEED_LONG = 4;    // a line of FQ_MAX_TOKEN bytes or more ("bufio.Scanner: token too long")
constexpr uint32_t FQ_NEED_BADID = 8;   // a header line that does not begin with '@' (seqio.go:38-40)

// Scalars of one parsed block (device memory; copied to the host behind the kernels).  Offsets are bytes from the start of
// the block's raw buffer: [start, end) is what was parsed — the previous block's tail in the porch, then this block's bytes.
struct FqState {
    uint32_t start, end;
    uint32_t n_lines;        // '\n' terminated lines in [start, end)
    uint32_t end_state;      // slot state behind the last line (0..3)
    uint32_t last_complete;  // lines up to and including the last one that completed a record (0: none did)
    uint32_t n_seq;          // sequence lines (slot 1, non-empty) — the last one is `pending` if its record is not complete
    uint32_t pending;
    uint32_t need_host;      // FQ_NEED_* (0: the device's result stands)
    unsigned long long seq_bytes;   // bases of the n_seq sequence lines
    uint32_t min_len, max_len;      // over the n_seq - pending reads of the block
    uint32_t tail_start, tail_len;  // [tail_start, end): behind the last completed record — the next block parses it again
    uint32_t tail_lines;            // whole lines inside the tail
    uint32_t pending_len;           // bases of the pending sequence line (they are in seq_bytes)
};

// index arrays of the parser (one set per context: the kernels of consecutive blocks run in order on one stream)
struct FqBuffers {
    uint32_t porch = 0;          // bytes reserved in front of a block's own bytes for the previous block's tail
    uint32_t line_cap = 0;       // lines the index holds
    uint32_t read_cap = 0;       // reads per block the output holds
    uint64_t bytes_cap = 0;      // bases per block the output holds
    uint32_t *wgcnt = nullptr, *line_end = nullptr, *linfo = nullptr, *wgmap = nullptr, *wgseq = nullptr, *src_out = nullptr;
    uint8_t *lmap = nullptr, *wgstate = nullptr;
    unsigned long long *wgbytes = nullptr;
};

// Parse raw[porch - tail .. porch + len) on stream s: `prev_raw` / `prev_state` (null for the first block of a stream) name
// the block whose tail is taken over.  Results: *state, off_out[0 .. n_seq] (exclusive byte offsets), bases_out.
hipError_t launch_fq_parse(hipStream_t s, const FqBuffers &B, const uint8_t *prev_raw, const FqState *prev_state, uint8_t *raw,
                           FqState *state, uint32_t len, uint64_t *off_out, uint8_t *bases_out);

// ---- FASTA (sketch.go:102-135) on the device --------------------------------------------------------------------------
// A '>' line closes the record in progress and opens the next; every other line is sequence, appended; the first EMPTY line
// ends the parsing.  Sequences are unbounded (a chromosome spans hundreds of blocks), lines are not: a block's sequence
// lines are compacted to the end of an accumulation buffer that outlives the block, header lines leave the position they
// stood at (the record offsets).  Nothing here needs the host parser: an empty line, a line of 64 KiB or more and the
// unterminated tail are reported in the block's scalars and the host acts on them in stream order.
constexpr uint32_t FA_NONE = 0xffffffffu;
struct FaState {
    uint32_t start, end;
    uint32_t n_lines;        // '\n' terminated lines in [start, end), at most the index's capacity (see FaBuffers::line_cap)
    uint32_t need_host;      // (FQ_NEED_LINES when the count was clamped: then first_empty is set, and nothing behind it counts)
    uint32_t first_empty;    // index of the first empty line (FA_NONE: none) — sketch.go:103-105: break
    uint32_t long_line;      // index of the first line of FQ_MAX_TOKEN bytes or more (FA_NONE: none)
    uint32_t n_hdr;          // '>' lines in front of the first of those two
    uint32_t tail_start, tail_len;   // the unterminated line at the block's end: the next block parses it again
    uint32_t min_len, max_len;       // over the records that begin AND end at a header of this block (n_hdr - 1 of them)
    uint32_t first_hdr, last_hdr;    // sequence bytes of the block in front of its first / its last header
    unsigned long long seq_bytes;    // sequence bytes of the block (in front of the first event)
};
// One set of index arrays per block in flight (two: a block is indexed while the one before it is placed).
struct FaBuffers {
    uint32_t porch = 0;
    // Lines the index holds: (64 KiB + block) / 2 + 2.  A non-empty line takes two bytes at least, so a block with more lines
    // than that has an empty one among the first line_cap — where the parsing ends anyway: clamping the index loses nothing.
    uint32_t line_cap = 0;
    uint32_t *wgcnt = nullptr, *line_end = nullptr, *linfo = nullptr, *ldst = nullptr, *wghdr = nullptr, *hrel = nullptr;
    unsigned long long *wgbytes = nullptr;
};
// A block in two steps, both on stream s.
//   index : raw[porch - tail .. porch + len) -> the line index, every sequence line's place in the block's own sequence bytes
//           (ldst), every header's (hrel), the block's scalars (*state).  Needs nothing from the blocks before but the tail.
//   place : the sequence lines to acc + out_base .., the headers' positions (out_base + hrel) to rec_off[0 .. n_hdr) — queued
//           once the host knows where the blocks before left the accumulation buffer (their scalars).
hipError_t launch_fa_index(hipStream_t s, const FaBuffers &B, const uint8_t *prev_raw, const FaState *prev_state, uint8_t *raw,
                           FaState *state, uint32_t len);
hipError_t launch_fa_place(hipStream_t s, const FaBuffers &B, const uint8_t *raw, const FaState *state, uint8_t *acc, uint64_t out_base,
                           uint64_t *rec_off);

}  // namespace hulk
