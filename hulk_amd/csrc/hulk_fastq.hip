// hulk_fastq.hip — FASTQ text -> reads ON THE DEVICE: the line machine of FastqHandler.Run (src/pipeline/sketch.go:99-161)
// over bufio.Scanner's ScanLines tokens (sketch.go:40-79; '\n' delimited, one trailing '\r' dropped), as a chain of data-
// parallel kernels over a raw block of file bytes that the host only read() into pinned memory and copied over PCIe.
//
//   k_fq_tail_in    the bytes behind the last completed record of the previous block move in front of this block's bytes
//                   (the "porch" of the raw buffer): every block is parsed from a record boundary, slot state 0
//   k_fq_count / k_fq_scan_u32 / k_fq_lines      newline index: 16 bytes per thread, SWAR byte compare, counts -> prefix ->
//                   line_end[i] = offset of the i-th '\n'
//   k_fq_class / k_fq_scan_maps                  per line: length (CR dropped), first byte, and the line's transition of the
//                   four-slot machine as a map {0..3} -> {0..3} in one byte (a non-empty line: s -> s+1 mod 4; an empty line:
//                   skipped in slots 0..2, completes the record in slot 3 — sketch.go:137-159); the maps compose
//                   associatively, so the slot every line meets is a prefix scan, empty lines and all
//   k_fq_flags / k_fq_scan_cnt / k_fq_emit       sequence lines (non-empty, slot 1) -> read index and byte offset by prefix
//                   sums; the last completing line (slot 3) bounds the block: a sequence behind it is not a read yet and
//                   stays in the tail
//   k_fq_copy       the sequences, back to back: the (bases, offsets) layout hulk_add_reads_device takes
//
// --fasta (sketch.go:102-135) is the second half of this file (k_fa_*): same newline index, no slot machine — sequence lines are
// compacted to the end of an accumulation buffer that outlives the block, header lines become the record offsets.
//
// What the device does NOT decide (FASTQ): anything the reference turns into an error or that outgrows the fixed buffers — a header
// line that does not begin with '@' (seqio.go:38-40), a line of 64 KiB or more (bufio.Scanner: token too long), more lines
// than the index holds, a tail longer than the porch.  The block is then flagged `need_host` and the host parser
// (hulk_ingest.hip, the comparator of tools/fuzz_ingest.py) takes the stream over from the last record boundary, with the
// reference's messages and their order.
#include "hulk_fastq.h"

#include <algorithm>

#include "hulk_device.h"

namespace hulk {
namespace {

constexpr uint32_t MAP_ID = 0xE4, MAP_NONEMPTY = 0x39, MAP_EMPTY = 0x24;   // x -> (m >> 2x) & 3
__device__ __forceinline__ uint32_t map_apply(uint32_t m, uint32_t x) { return (m >> (2u * x)) & 3u; }
// (a then b)
__device__ __forceinline__ uint32_t map_then(uint32_t a, uint32_t b) {
    return map_apply(b, a & 3u) | map_apply(b, (a >> 2) & 3u) << 2 | map_apply(b, (a >> 4) & 3u) << 4 | map_apply(b, (a >> 6) & 3u) << 6;
}

// 0x80 in every byte of w that equals c (exact, no carries between bytes)
__device__ __forceinline__ uint32_t bytes_eq(uint32_t w, uint32_t c4) {
    const uint32_t x = w ^ c4;
    return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
}
// bit i = byte i of the thread's 16 bytes is '\n' and lies in [lo, hi)
__device__ __forceinline__ uint32_t newline_mask(const uint8_t *raw, uint32_t addr, uint32_t lo, uint32_t hi) {
    if (addr + 16u <= lo || addr >= hi) return 0u;
    const uint4 v = *(const uint4 *)(raw + addr);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) m |= ((((bytes_eq(w[i], 0x0A0A0A0Au) >> 7) * 0x10204080u) >> 28) & 0xFu) << (4 * i);
    if (addr < lo) m &= ~((1u << (lo - addr)) - 1u);
    if (addr + 16u > hi) m &= (1u << (hi - addr)) - 1u;
    return m;
}

constexpr int FQ_T = 256;                      // threads per workgroup of the index kernels: small ones find room at once beside the binning
                                               // kernels, which leave a few wave slots per CU free (1024-thread ones waited for half a CU)
constexpr uint32_t FQ_CHUNK = FQ_T * 16;       // bytes per workgroup

// workgroup-wide exclusive prefix sum of one uint32 per thread (T threads); returns the workgroup total through `total`
template <int T = FQ_T>
__device__ __forceinline__ uint32_t wg_excl_u32(uint32_t v, uint32_t *lds /* [T / 64 + 1] */, uint32_t &total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    if (lane == 63) lds[wid] = incl;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (int i = 0; i < T / 64; i++) { const uint32_t t = lds[i]; lds[i] = run; run += t; } lds[T / 64] = run; }
    __syncthreads();
    total = lds[T / 64];
    const uint32_t r = lds[wid] + incl - v;
    __syncthreads();
    return r;
}
template <int T = FQ_T>
__device__ __forceinline__ unsigned long long wg_excl_u64(unsigned long long v, unsigned long long *lds, unsigned long long &total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned long long incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    if (lane == 63) lds[wid] = incl;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long run = 0; for (int i = 0; i < T / 64; i++) { const unsigned long long t = lds[i]; lds[i] = run; run += t; } lds[T / 64] = run; }
    __syncthreads();
    total = lds[T / 64];
    const unsigned long long r = lds[wid] + incl - v;
    __syncthreads();
    return r;
}
// workgroup-wide exclusive prefix of transition maps (composition in line order)
__device__ __forceinline__ uint32_t wg_excl_map(uint32_t m, uint32_t *lds, uint32_t &total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t incl = m;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= d) incl = map_then(o, incl); }
    uint32_t excl = __shfl_up(incl, 1);
    if (lane == 0) excl = MAP_ID;
    if (lane == 63) lds[wid] = incl;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = MAP_ID; for (int i = 0; i < FQ_T / 64; i++) { const uint32_t t = lds[i]; lds[i] = run; run = map_then(run, t); } lds[FQ_T / 64] = run; }
    __syncthreads();
    total = lds[FQ_T / 64];
    const uint32_t r = map_then(lds[wid], excl);
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void k_fq_tail_in(const uint8_t *__restrict__ prev_raw, const FqState *__restrict__ prev,
                                                    uint8_t *__restrict__ raw, FqState *st, uint32_t porch, uint32_t len) {
    const uint32_t tl = prev ? prev->tail_len : 0u;
    const bool fits = tl <= porch;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        FqState s{};
        s.start = fits ? porch - tl : porch;
        s.end = porch + len;
        s.need_host = fits ? 0u : FQ_NEED_TAIL;
        s.min_len = 0xffffffffu;
        s.last_complete = 0;
        *st = s;
    }
    if (!fits || !tl) return;
    const uint8_t *src = prev_raw + prev->tail_start;
    uint8_t *dst = raw + porch - tl;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < tl; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

template <class ST>                                            // FqState / FaState: start, end, n_lines, need_host
__global__ __launch_bounds__(FQ_T) void k_fq_count(const uint8_t *__restrict__ raw, const ST *__restrict__ st,
                                                   uint32_t *__restrict__ wgcnt) {
    __shared__ uint32_t red[FQ_T / 64];
    const uint32_t lo = st->start, hi = st->end, a0 = lo & ~15u;
    const uint32_t addr = a0 + blockIdx.x * FQ_CHUNK + threadIdx.x * 16u;
    if (a0 + blockIdx.x * FQ_CHUNK >= hi) { if (threadIdx.x == 0) wgcnt[blockIdx.x] = 0; return; }
    uint32_t c = (uint32_t)__popc(newline_mask(raw, addr, lo, hi));
    for (int off = 32; off; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t t = 0; for (int i = 0; i < FQ_T / 64; i++) t += red[i]; wgcnt[blockIdx.x] = t; }
}

// exclusive scan of the n workgroup counts in place (8 * FQ_T per trip); the total goes to n_lines (clamped to cap -> need_host)
template <class ST>
__global__ __launch_bounds__(FQ_T) void k_fq_scan_u32(uint32_t *__restrict__ v, uint32_t n, ST *st, uint32_t cap) {
    __shared__ uint32_t lds[FQ_T / 64 + 1];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += 8u * FQ_T) {
        uint32_t mine[8], sum = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { const uint32_t at = base + threadIdx.x * 8 + i; mine[i] = at < n ? v[at] : 0u; sum += mine[i]; }
        uint32_t total;
        uint32_t run = carry + wg_excl_u32(sum, lds, total);
#pragma unroll
        for (int i = 0; i < 8; i++) { const uint32_t at = base + threadIdx.x * 8 + i; if (at < n) v[at] = run; run += mine[i]; }
        carry += total;
    }
    if (threadIdx.x == 0) {
        if (carry > cap) { st->need_host |= FQ_NEED_LINES; carry = cap; }
        st->n_lines = carry;
    }
}

template <class ST>
__global__ __launch_bounds__(FQ_T) void k_fq_lines(const uint8_t *__restrict__ raw, const ST *__restrict__ st,
                                                   const uint32_t *__restrict__ wgbase, uint32_t *__restrict__ line_end, uint32_t cap) {
    __shared__ uint32_t lds[FQ_T / 64 + 1];
    const uint32_t lo = st->start, hi = st->end, a0 = lo & ~15u;
    if (a0 + blockIdx.x * FQ_CHUNK >= hi) return;
    const uint32_t addr = a0 + blockIdx.x * FQ_CHUNK + threadIdx.x * 16u;
    uint32_t m = newline_mask(raw, addr, lo, hi);
    uint32_t total;
    uint32_t at = wgbase[blockIdx.x] + wg_excl_u32((uint32_t)__popc(m), lds, total);
    while (m) {
        const uint32_t b = (uint32_t)__ffs((int)m) - 1u;
        m &= m - 1u;
        if (at < cap) line_end[at] = addr + b;
        at++;
    }
}

// per line: length, first byte, transition map; exclusive prefix of the maps inside the workgroup, the workgroup's own map
__global__ __launch_bounds__(FQ_T) void k_fq_class(const uint8_t *__restrict__ raw, FqState *st, const uint32_t *__restrict__ line_end,
                                                   uint32_t *__restrict__ linfo, uint8_t *__restrict__ lmap, uint32_t *__restrict__ wgmap) {
    __shared__ uint32_t lds[FQ_T / 64 + 1];
    const uint32_t NL = st->n_lines;
    if (blockIdx.x * FQ_T >= NL) { if (threadIdx.x == 0) wgmap[blockIdx.x] = MAP_ID; return; }
    const uint32_t i = blockIdx.x * FQ_T + threadIdx.x;
    uint32_t m = MAP_ID;
    if (i < NL) {
        const uint32_t b = i ? line_end[i - 1] + 1u : st->start, e = line_end[i];
        const uint32_t rawlen = e - b;
        uint32_t L = rawlen;
        if (L && raw[e - 1] == '\r') L--;                           // ScanLines' dropCR
        const uint32_t first = L ? raw[b] : 0u;
        if (rawlen >= FQ_MAX_TOKEN) atomicOr(&st->need_host, FQ_NEED_LONG);
        linfo[i] = (L < 0xffffffu ? L : 0xffffffu) | first << 24;
        m = L ? MAP_NONEMPTY : MAP_EMPTY;
    }
    uint32_t total;
    const uint32_t ex = wg_excl_map(m, lds, total);
    if (i < NL) lmap[i] = (uint8_t)ex;
    if (threadIdx.x == 0) wgmap[blockIdx.x] = total;
}

// slot state in front of every workgroup of lines (the stream enters a block in state 0), and after the last line
__global__ __launch_bounds__(FQ_T) void k_fq_scan_maps(const uint32_t *__restrict__ wgmap, uint32_t n, uint8_t *__restrict__ wgstate,
                                                       FqState *st) {
    __shared__ uint32_t lds[FQ_T / 64 + 1];
    uint32_t carry = MAP_ID;                                        // composition of everything in front of this trip
    for (uint32_t base = 0; base < n || base == 0; base += 8u * FQ_T) {
        uint32_t mine[8], mm = MAP_ID;
#pragma unroll
        for (int i = 0; i < 8; i++) { const uint32_t at = base + threadIdx.x * 8 + i; mine[i] = at < n ? wgmap[at] : MAP_ID; mm = map_then(mm, mine[i]); }
        uint32_t total;
        uint32_t run = map_then(carry, wg_excl_map(mm, lds, total));
#pragma unroll
        for (int i = 0; i < 8; i++) { const uint32_t at = base + threadIdx.x * 8 + i; if (at < n) wgstate[at] = (uint8_t)map_apply(run, 0u); run = map_then(run, mine[i]); }
        carry = map_then(carry, total);
    }
    if (threadIdx.x == 0) st->end_state = map_apply(carry, 0u);
}

// sequence lines and completing lines; per workgroup: number of sequence lines and their bytes
__global__ __launch_bounds__(FQ_T) void k_fq_flags(FqState *st, const uint32_t *__restrict__ linfo, const uint8_t *__restrict__ lmap,
                                                   const uint8_t *__restrict__ wgstate, uint32_t *__restrict__ wgseq,
                                                   unsigned long long *__restrict__ wgbytes) {
    __shared__ uint32_t rc[FQ_T / 64]; __shared__ unsigned long long rb[FQ_T / 64];
    const uint32_t NL = st->n_lines;
    if (blockIdx.x * FQ_T >= NL) { if (threadIdx.x == 0) { wgseq[blockIdx.x] = 0; wgbytes[blockIdx.x] = 0; } return; }
    const uint32_t i = blockIdx.x * FQ_T + threadIdx.x;
    uint32_t c = 0; unsigned long long by = 0;
    if (i < NL) {
        const uint32_t info = linfo[i], L = info & 0xffffffu, first = info >> 24;
        const uint32_t s = map_apply(lmap[i], wgstate[blockIdx.x]);
        if (s == 1u && L) { c = 1; by = L; }
        if (s == 0u && L && first != '@') atomicOr(&st->need_host, FQ_NEED_BADID);     // seqio.go:38-40: the host words the message
        if (s == 3u) atomicMax(&st->last_complete, i + 1u);                            // any line in slot 3 completes the record
    }
    for (int off = 32; off; off >>= 1) { c += __shfl_xor(c, off); by += __shfl_xor(by, off); }
    if ((threadIdx.x & 63) == 0) { rc[threadIdx.x >> 6] = c; rb[threadIdx.x >> 6] = by; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tc = 0; unsigned long long tb = 0;
        for (int x = 0; x < FQ_T / 64; x++) { tc += rc[x]; tb += rb[x]; }
        wgseq[blockIdx.x] = tc; wgbytes[blockIdx.x] = tb;
    }
}

// exclusive scans of the workgroup sums (8 * FQ_T per trip); the block's scalars
__global__ __launch_bounds__(FQ_T) void k_fq_scan_cnt(uint32_t *__restrict__ wgseq, unsigned long long *__restrict__ wgbytes, uint32_t n,
                                                      FqState *st, const uint32_t *__restrict__ line_end, uint64_t *__restrict__ off_out,
                                                      uint32_t read_cap, uint64_t bytes_cap) {
    __shared__ uint32_t lds[FQ_T / 64 + 1]; __shared__ unsigned long long ldb[FQ_T / 64 + 1];
    uint32_t tc = 0; unsigned long long tb = 0;                       // running totals
    for (uint32_t base = 0; base < n || base == 0; base += 8u * FQ_T) {
        uint32_t mc[8], sc = 0; unsigned long long mb[8], sb = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t at = base + threadIdx.x * 8 + i;
            mc[i] = at < n ? wgseq[at] : 0u; mb[i] = at < n ? wgbytes[at] : 0ull; sc += mc[i]; sb += mb[i];
        }
        uint32_t t1; unsigned long long t2;
        uint32_t rcn = tc + wg_excl_u32(sc, lds, t1);
        unsigned long long rbn = tb + wg_excl_u64(sb, ldb, t2);
#pragma unroll
        for (int i = 0; i < 8; i++) { const uint32_t at = base + threadIdx.x * 8 + i; if (at < n) { wgseq[at] = rcn; wgbytes[at] = rbn; } rcn += mc[i]; rbn += mb[i]; }
        tc += t1; tb += t2;
    }
    if (threadIdx.x == 0) {
        const uint32_t lc = st->last_complete;                      // number of lines up to and including the last completing one
        st->n_seq = tc;
        st->seq_bytes = tb;
        // a sequence line behind the last completing line is the record in progress: it stays in the tail
        st->pending = (st->end_state >= 2u && tc) ? 1u : 0u;
        st->tail_start = lc ? line_end[lc - 1] + 1u : st->start;
        st->tail_len = st->end - st->tail_start;
        st->tail_lines = st->n_lines - lc;
        if (tc > read_cap || tb > bytes_cap) st->need_host |= FQ_NEED_LINES;
        if (tc <= read_cap) off_out[tc] = tb;
    }
}

__global__ __launch_bounds__(FQ_T) void k_fq_emit(FqState *st, const uint32_t *__restrict__ line_end, const uint32_t *__restrict__ linfo,
                                                  const uint8_t *__restrict__ lmap, const uint8_t *__restrict__ wgstate,
                                                  const uint32_t *__restrict__ wgseq, const unsigned long long *__restrict__ wgbytes,
                                                  uint64_t *__restrict__ off_out, uint32_t *__restrict__ src_out, uint32_t read_cap) {
    __shared__ uint32_t lds[FQ_T / 64 + 1]; __shared__ unsigned long long ldb[FQ_T / 64 + 1];
    const uint32_t NL = st->n_lines;
    if (blockIdx.x * FQ_T >= NL) return;
    const uint32_t i = blockIdx.x * FQ_T + threadIdx.x;
    uint32_t c = 0, L = 0, b = 0;
    if (i < NL) {
        const uint32_t info = linfo[i];
        L = info & 0xffffffu;
        const uint32_t s = map_apply(lmap[i], wgstate[blockIdx.x]);
        if (s == 1u && L) { c = 1; b = i ? line_end[i - 1] + 1u : st->start; }
    }
    uint32_t tc; unsigned long long tb;
    const uint32_t idx = wgseq[blockIdx.x] + wg_excl_u32(c, lds, tc);
    const unsigned long long at = wgbytes[blockIdx.x] + wg_excl_u64(c ? (unsigned long long)L : 0ull, ldb, tb);
    if (c && idx < read_cap) {
        off_out[idx] = at; src_out[idx] = b;
        const bool pending = st->pending && idx + 1u == st->n_seq;
        if (!pending) { atomicMin(&st->min_len, L); atomicMax(&st->max_len, L); }
        else st->pending_len = L;
    }
}

__global__ __launch_bounds__(256) void k_fq_copy(const uint8_t *__restrict__ raw, const FqState *__restrict__ st,
                                                 const uint64_t *__restrict__ off_out, const uint32_t *__restrict__ src_out,
                                                 uint8_t *__restrict__ bases_out, uint32_t read_cap, uint64_t bytes_cap) {
    const uint32_t n = st->n_seq < read_cap ? st->n_seq : read_cap;
    if (st->seq_bytes > bytes_cap) return;
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t r = wave; r < n; r += nw) {
        const uint64_t o = off_out[r];
        const uint32_t L = (uint32_t)(off_out[r + 1] - o), s = src_out[r];
        for (uint32_t x = lane; x < L; x += 64) bases_out[o + x] = raw[s + x];
    }
}


// ------------------------------------------------------------------------------------------
// FASTA (hulk_fastq.h).  A block in two steps — index (everything that depends on the block's own bytes only) and place (the
// copy into the accumulation buffer, once the host knows where the blocks before left it):
//   k_fq_count / k_fq_scan_u32 / k_fq_lines      the newline index: the FASTQ parser's kernels
//   k_fa_class   per line: length (CR dropped), first byte; the first empty line and the first line of 64 KiB or more (atomicMin)
//   k_fa_flags / k_fa_scan    header lines and sequence bytes in front of the first of those, per workgroup -> prefix sums
//   k_fa_emit    a sequence line's destination and a header's position, both in bytes from the block's first sequence byte
//   k_fa_lens    shortest / longest record between two headers of the block, the block's first / last header position
//   -- place --
//   k_fa_copy    the sequence lines to the end of the accumulation buffer, 8 lines per wave and round
//   k_fa_recs    the headers' positions in the accumulation buffer -> the record offsets
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fa_begin(const uint8_t *__restrict__ prev_raw, const FaState *__restrict__ prev,
                                                  uint8_t *__restrict__ raw, FaState *st, uint32_t porch, uint32_t len) {
    const uint32_t tl = prev ? prev->tail_len : 0u;
    const bool fits = tl <= porch;               // (the host ends the run at a tail of FQ_MAX_TOKEN bytes: it always fits)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        FaState s{};
        s.start = fits ? porch - tl : porch;
        s.end = porch + len;
        s.need_host = fits ? 0u : FQ_NEED_TAIL;
        s.first_empty = FA_NONE; s.long_line = FA_NONE;
        s.min_len = FA_NONE; s.max_len = 0;
        *st = s;
    }
    if (!fits || !tl) return;
    const uint8_t *src = prev_raw + prev->tail_start;
    uint8_t *dst = raw + porch - tl;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < tl; i += gridDim.x * blockDim.x) dst[i] = src[i];
}


// The chain runs beside the context's binning kernels (k_long_tile fills every CU with 256-thread workgroups and leaves a few
// wave slots free): 256-thread workgroups (FQ_T) find room at once — a 1024-thread workgroup waited until half a CU happened to
// be free (measured: the one-workgroup scans went from 15 to 200 us).
constexpr int FA_T = FQ_T;
constexpr uint32_t FA_CHUNK = FQ_CHUNK;

__global__ __launch_bounds__(FA_T) void k_fa_class(const uint8_t *__restrict__ raw, FaState *st, const uint32_t *__restrict__ line_end,
                                                   uint32_t *__restrict__ linfo) {
    const uint32_t NL = st->n_lines;
    // (the grid is sized for lines of 16 bytes; a block of shorter ones takes more trips)
    for (uint32_t i = blockIdx.x * FA_T + threadIdx.x; i < NL; i += gridDim.x * FA_T) {
        const uint32_t b = i ? line_end[i - 1] + 1u : st->start, e = line_end[i];
        const uint32_t rawlen = e - b;
        uint32_t L = rawlen;
        if (L && raw[e - 1] == '\r') L--;                               // ScanLines' dropCR
        const uint32_t first = L ? raw[b] : 0u;
        if (rawlen >= FQ_MAX_TOKEN) atomicMin(&st->long_line, i);
        if (L == 0) atomicMin(&st->first_empty, i);
        linfo[i] = (L < 0xffffffu ? L : 0xffffffu) | first << 24;
    }
}

// lines in front of the first event (empty line / line too long) count
__device__ __forceinline__ uint32_t fa_live_lines(const FaState *st) {
    uint32_t ev = st->n_lines;
    if (st->first_empty < ev) ev = st->first_empty;
    if (st->long_line < ev) ev = st->long_line;
    return ev;
}

__global__ __launch_bounds__(FA_T) void k_fa_flags(const FaState *st, const uint32_t *__restrict__ linfo, uint32_t *__restrict__ wghdr,
                                                   unsigned long long *__restrict__ wgbytes) {
    __shared__ uint32_t rc[FA_T / 64]; __shared__ unsigned long long rb[FA_T / 64];
    const uint32_t ev = fa_live_lines(st);
    for (uint32_t blk = blockIdx.x; blk * FA_T < ev; blk += gridDim.x) {      // chunk `blk` of FA_T lines (k_fa_scan reads no chunk behind ev)
        const uint32_t i = blk * FA_T + threadIdx.x;
        uint32_t c = 0; unsigned long long by = 0;
        if (i < ev) {
            const uint32_t info = linfo[i];
            if ((info >> 24) == '>') c = 1; else by = info & 0xffffffu;
        }
        for (int off = 32; off; off >>= 1) { c += __shfl_xor(c, off); by += __shfl_xor(by, off); }
        if ((threadIdx.x & 63) == 0) { rc[threadIdx.x >> 6] = c; rb[threadIdx.x >> 6] = by; }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tc = 0; unsigned long long tb = 0;
            for (int x = 0; x < FA_T / 64; x++) { tc += rc[x]; tb += rb[x]; }
            wghdr[blk] = tc; wgbytes[blk] = tb;
        }
        __syncthreads();
    }
}

// exclusive scans of the workgroup sums (16 * FA_T of them per trip); the block's scalars
__global__ __launch_bounds__(FA_T) void k_fa_scan(uint32_t *__restrict__ wghdr, unsigned long long *__restrict__ wgbytes, uint32_t n,
                                                  FaState *st, const uint32_t *__restrict__ line_end) {
    __shared__ uint32_t lds[FA_T / 64 + 1]; __shared__ unsigned long long ldb[FA_T / 64 + 1];
    { const uint32_t used = (fa_live_lines(st) + FA_T - 1) / FA_T; if (used < n) n = used; }     // (chunks behind the last live line were not written)
    uint32_t carry_c = 0; unsigned long long carry_b = 0;
    for (uint32_t base = 0; base < n || base == 0; base += 16u * FA_T) {
        uint32_t mc[16], sc = 0; unsigned long long mb[16], sb = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t at = base + threadIdx.x * 16 + i;
            mc[i] = at < n ? wghdr[at] : 0u; mb[i] = at < n ? wgbytes[at] : 0ull; sc += mc[i]; sb += mb[i];
        }
        uint32_t tc; unsigned long long tb;
        uint32_t rcn = carry_c + wg_excl_u32<FA_T>(sc, lds, tc);
        unsigned long long rbn = carry_b + wg_excl_u64<FA_T>(sb, ldb, tb);
#pragma unroll
        for (int i = 0; i < 16; i++) { const uint32_t at = base + threadIdx.x * 16 + i; if (at < n) { wghdr[at] = rcn; wgbytes[at] = rbn; } rcn += mc[i]; rbn += mb[i]; }
        carry_c += tc; carry_b += tb;
    }
    if (threadIdx.x == 0) {
        st->n_hdr = carry_c;
        st->seq_bytes = carry_b;
        const uint32_t NL = st->n_lines;
        st->tail_start = NL ? line_end[NL - 1] + 1u : st->start;
        st->tail_len = st->end - st->tail_start;
    }
}

__global__ __launch_bounds__(FA_T) void k_fa_emit(const FaState *st, const uint32_t *__restrict__ linfo, const uint32_t *__restrict__ wghdr,
                                                  const unsigned long long *__restrict__ wgbytes, uint32_t *__restrict__ ldst,
                                                  uint32_t *__restrict__ hrel) {
    __shared__ uint32_t lds[FA_T / 64 + 1]; __shared__ unsigned long long ldb[FA_T / 64 + 1];
    const uint32_t ev = fa_live_lines(st);
    for (uint32_t blk = blockIdx.x; blk * FA_T < ev; blk += gridDim.x) {
        const uint32_t i = blk * FA_T + threadIdx.x;
        uint32_t c = 0, L = 0;
        if (i < ev) {
            const uint32_t info = linfo[i];
            if ((info >> 24) == '>') c = 1; else L = info & 0xffffffu;
        }
        uint32_t tc; unsigned long long tb;
        const uint32_t hidx = wghdr[blk] + wg_excl_u32<FA_T>(c, lds, tc);
        const unsigned long long at = wgbytes[blk] + wg_excl_u64<FA_T>((unsigned long long)L, ldb, tb);
        if (i < ev) {                                               // (a block's sequence bytes are fewer than its raw bytes: 32 bits)
            if (c) { hrel[hidx] = (uint32_t)at; ldst[i] = FA_NONE; }
            else ldst[i] = (uint32_t)at;
        }
    }
}

__global__ __launch_bounds__(256) void k_fa_copy(const uint8_t *__restrict__ raw, const FaState *__restrict__ st,
                                                 const uint32_t *__restrict__ line_end, const uint32_t *__restrict__ linfo,
                                                 const uint32_t *__restrict__ ldst, uint8_t *__restrict__ out) {
    const uint32_t ev = fa_live_lines(st);
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    // 8 consecutive lines per wave and round: lanes 0..7 fetch a line's (source, destination, length) each; the first 128 bytes of all
    // eight are loaded before any is stored (a line is 60-80 bytes: one dependent round trip per round, not per line), what is left
    // of a longer line follows 64 bytes at a time
    constexpr uint32_t R = 8;
    for (uint32_t r0 = wave * R; r0 < ev; r0 += nw * R) {
        const uint32_t r = r0 + (lane & (R - 1));
        uint32_t src = 0, dst = FA_NONE, L = 0;
        if (r < ev) { dst = ldst[r]; L = linfo[r] & 0xffffffu; src = r ? line_end[r - 1] + 1u : st->start; }
        if (dst == FA_NONE) L = 0;                                  // (a header line, or no line)
        uint8_t v0[R], v1[R]; uint32_t s_[R], d_[R], n_[R];
#pragma unroll
        for (uint32_t x = 0; x < R; x++) {
            s_[x] = (uint32_t)__shfl((int)src, (int)x); d_[x] = (uint32_t)__shfl((int)dst, (int)x); n_[x] = (uint32_t)__shfl((int)L, (int)x);
            v0[x] = lane < n_[x] ? raw[s_[x] + lane] : (uint8_t)0;
            v1[x] = lane + 64u < n_[x] ? raw[s_[x] + lane + 64u] : (uint8_t)0;
        }
#pragma unroll
        for (uint32_t x = 0; x < R; x++) {
            if (lane < n_[x]) out[(size_t)d_[x] + lane] = v0[x];
            if (lane + 64u < n_[x]) out[(size_t)d_[x] + lane + 64u] = v1[x];
        }
#pragma unroll
        for (uint32_t x = 0; x < R; x++)
            for (uint32_t y = 128u + lane; y < n_[x]; y += 64u) out[(size_t)d_[x] + y] = raw[s_[x] + y];
    }
}

__global__ __launch_bounds__(256) void k_fa_lens(FaState *st, const uint32_t *__restrict__ hrel) {
    const uint32_t n = st->n_hdr;
    uint32_t mn = FA_NONE, mx = 0;
    for (uint32_t h = 1u + blockIdx.x * blockDim.x + threadIdx.x; h < n; h += gridDim.x * blockDim.x) {
        const uint32_t d = hrel[h] - hrel[h - 1];
        mn = d < mn ? d : mn; mx = d > mx ? d : mx;
    }
    for (int off = 32; off; off >>= 1) { const uint32_t a = __shfl_xor(mn, off), b = __shfl_xor(mx, off); mn = a < mn ? a : mn; mx = b > mx ? b : mx; }
    if ((threadIdx.x & 63) == 0) { if (mn != FA_NONE) atomicMin(&st->min_len, mn); if (mx) atomicMax(&st->max_len, mx); }
    if (blockIdx.x == 0 && threadIdx.x == 0 && n) { st->first_hdr = hrel[0]; st->last_hdr = hrel[n - 1]; }
}

// the block's headers -> record offsets of the accumulation buffer
__global__ __launch_bounds__(256) void k_fa_recs(const FaState *__restrict__ st, const uint32_t *__restrict__ hrel, uint64_t *__restrict__ rec_off,
                                                 uint64_t out_base) {
    const uint32_t n = st->n_hdr;
    for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h < n; h += gridDim.x * blockDim.x) rec_off[h] = out_base + hrel[h];
}

}  // namespace

hipError_t launch_fq_parse(hipStream_t s, const FqBuffers &B, const uint8_t *prev_raw, const FqState *prev_state, uint8_t *raw,
                           FqState *state, uint32_t len, uint64_t *off_out, uint8_t *bases_out) {
    const uint32_t span = B.porch + len + 16u;
    const uint32_t nchunks = (span + FQ_CHUNK - 1) / FQ_CHUNK;
    const uint32_t nlwg = (B.line_cap + FQ_T - 1) / FQ_T;
    hipLaunchKernelGGL(k_fq_tail_in, dim3(64), dim3(256), 0, s, prev_raw, prev_state, raw, state, B.porch, len);
    hipLaunchKernelGGL(k_fq_count<FqState>, dim3(nchunks), dim3(FQ_T), 0, s, raw, state, B.wgcnt);
    hipLaunchKernelGGL(k_fq_scan_u32<FqState>, dim3(1), dim3(FQ_T), 0, s, B.wgcnt, nchunks, state, B.line_cap);
    hipLaunchKernelGGL(k_fq_lines<FqState>, dim3(nchunks), dim3(FQ_T), 0, s, raw, state, B.wgcnt, B.line_end, B.line_cap);
    hipLaunchKernelGGL(k_fq_class, dim3(nlwg), dim3(FQ_T), 0, s, raw, state, B.line_end, B.linfo, B.lmap, B.wgmap);
    hipLaunchKernelGGL(k_fq_scan_maps, dim3(1), dim3(FQ_T), 0, s, B.wgmap, nlwg, B.wgstate, state);
    hipLaunchKernelGGL(k_fq_flags, dim3(nlwg), dim3(FQ_T), 0, s, state, B.linfo, B.lmap, B.wgstate, B.wgseq, B.wgbytes);
    hipLaunchKernelGGL(k_fq_scan_cnt, dim3(1), dim3(FQ_T), 0, s, B.wgseq, B.wgbytes, nlwg, state, B.line_end, off_out, B.read_cap, B.bytes_cap);
    hipLaunchKernelGGL(k_fq_emit, dim3(nlwg), dim3(FQ_T), 0, s, state, B.line_end, B.linfo, B.lmap, B.wgstate, B.wgseq, B.wgbytes, off_out,
                       B.src_out, B.read_cap);
    hipLaunchKernelGGL(k_fq_copy, dim3(2048), dim3(256), 0, s, raw, state, off_out, B.src_out, bases_out, B.read_cap, B.bytes_cap);
    return hipGetLastError();
}

hipError_t launch_fa_index(hipStream_t s, const FaBuffers &B, const uint8_t *prev_raw, const FaState *prev_state, uint8_t *raw,
                           FaState *state, uint32_t len) {
    const uint32_t span = B.porch + len + 16u;
    const uint32_t nchunks = (span + FA_CHUNK - 1) / FA_CHUNK;
    const uint32_t nlwg = (B.line_cap + FA_T - 1) / FA_T;
    hipLaunchKernelGGL(k_fa_begin, dim3(64), dim3(256), 0, s, prev_raw, prev_state, raw, state, B.porch, len);
    hipLaunchKernelGGL(k_fq_count<FaState>, dim3(nchunks), dim3(FQ_T), 0, s, raw, state, B.wgcnt);
    hipLaunchKernelGGL(k_fq_scan_u32<FaState>, dim3(1), dim3(FQ_T), 0, s, B.wgcnt, nchunks, state, B.line_cap);
    hipLaunchKernelGGL(k_fq_lines<FaState>, dim3(nchunks), dim3(FQ_T), 0, s, raw, state, B.wgcnt, B.line_end, B.line_cap);
    // a block of 60-byte lines has 1/30 of line_cap: the grids are sized for lines of 16 bytes, the kernels take more trips over shorter ones
    const uint32_t glw = std::min<uint32_t>(nlwg, (span / 16u + FA_T - 1) / FA_T + 1u);
    hipLaunchKernelGGL(k_fa_class, dim3(glw), dim3(FA_T), 0, s, raw, state, B.line_end, B.linfo);
    hipLaunchKernelGGL(k_fa_flags, dim3(glw), dim3(FA_T), 0, s, state, B.linfo, B.wghdr, B.wgbytes);
    hipLaunchKernelGGL(k_fa_scan, dim3(1), dim3(FA_T), 0, s, B.wghdr, B.wgbytes, nlwg, state, B.line_end);
    hipLaunchKernelGGL(k_fa_emit, dim3(glw), dim3(FA_T), 0, s, state, B.linfo, B.wghdr, B.wgbytes, B.ldst, B.hrel);
    hipLaunchKernelGGL(k_fa_lens, dim3(64), dim3(256), 0, s, state, B.hrel);
    return hipGetLastError();
}

hipError_t launch_fa_place(hipStream_t s, const FaBuffers &B, const uint8_t *raw, const FaState *state, uint8_t *acc, uint64_t out_base,
                           uint64_t *rec_off) {
    hipLaunchKernelGGL(k_fa_copy, dim3(2048), dim3(256), 0, s, raw, state, B.line_end, B.linfo, B.ldst, acc + out_base);
    hipLaunchKernelGGL(k_fa_recs, dim3(64), dim3(256), 0, s, state, B.hrel, rec_off, out_base);
    return hipGetLastError();
}

}  // namespace hulk
