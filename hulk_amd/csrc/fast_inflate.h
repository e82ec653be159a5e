// fast_inflate.h — a DEFLATE (RFC 1951) decoder for the ingest path, written for throughput on one core.
//
// zlib's inflate delivers ~0.5 GB/s of text, which made `.gz` input two orders of magnitude slower than the plain
// FASTQ parser behind it.  This one keeps a 64-bit bit buffer that is refilled with one unaligned 8-byte load,
// decodes literal/length symbols through an 11-bit first-level table (second level for longer codes) whose
// entries already hold the extra-bit count and the base value, and copies matches eight bytes at a time.
// The caller owns the buffers: input is fed in pieces (`feed`), output goes to a caller-supplied window that is at
// least 32 KiB + slack larger than what one call may produce; everything is resumable at block and symbol
// granularity, so neither the compressed nor the inflated stream has to be in memory at once.
//
// Only what the gzip reader of the reference accepts is accepted (compress/gzip + compress/flate, Go 1.12):
// stored, fixed and dynamic blocks, distances up to 32 KiB, over-subscribed or incomplete code sets are errors
// (a distance code set with a single code is allowed, as in zlib and Go).
#pragma once
#include <stdint.h>
#include <string.h>

namespace hulk {
namespace inflate {

constexpr int LITLEN_BITS = 11, DIST_BITS = 8, PRE_BITS = 7;
constexpr int LITLEN_ENOUGH = 2342, DIST_ENOUGH = 402;     // table sizes sufficient for any code set (zlib's "enough")
constexpr size_t OUT_SLACK = 320;                          // bytes a fast-loop step may write past `out_limit`
constexpr size_t IN_SLACK = 24;                            // input bytes the fast loop wants ahead of every iteration (three 8-byte refills)

// table entry: bits 0-5 the bits to drop — code length, plus the extra bits for a length / distance symbol, so that ONE shift
// by the entry itself (a 6-bit count) moves the reader past the whole symbol and the extra bits' value is cut out of the old
// buffer off the critical path —, 8-12 extra-bit count, 13-15 kind, 16-31 value
enum Kind : uint32_t { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4 };
constexpr uint32_t KIND_SHIFT = 13, KIND_MASK = 7u << KIND_SHIFT;
static inline uint32_t mk(uint32_t len, uint32_t kind, uint32_t extra, uint32_t val) {
    return (len + (kind == K_LEN ? extra : 0u)) | (kind << KIND_SHIFT) | (extra << 8) | (val << 16);
}
static inline uint32_t drop_of(uint32_t e) { return e & 63u; }                                  // bits the symbol takes off the reader, extra bits included
static inline uint32_t codelen_of(uint32_t e) {                                                 // ... its Huffman code alone
    return ((e >> KIND_SHIFT) & 7u) == K_LEN ? (e & 63u) - ((e >> 8) & 31u) : (e & 63u);
}

static const uint16_t LEN_BASE[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
static const uint8_t LEN_EXTRA[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const uint16_t DIST_BASE[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
static const uint8_t DIST_EXTRA[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};

// Canonical Huffman decode table.  `what`: 0 = litlen (symbols 0..285), 1 = distance (0..29), 2 = precode (0..18).
// Returns false for an over-subscribed set, or an incomplete one (except: no codes at all, or a single
// 1-bit code, for the distance alphabet).
static inline bool build_table(const uint8_t *lens, int nsyms, int what, uint32_t *table, int table_bits, int table_cap) {
    int count[16] = {0};
    for (int i = 0; i < nsyms; i++) count[lens[i]]++;
    int left = 1, maxlen = 0, total = 0;
    for (int l = 1; l <= 15; l++) {
        left = (left << 1) - count[l];
        if (left < 0) return false;                            // over-subscribed
        if (count[l]) { maxlen = l; total += count[l]; }
    }
    const int prim = 1 << table_bits;
    if (total == 0) {                                          // no codes: every lookup is an error (legal for distances
        for (int i = 0; i < prim; i++) table[i] = mk(1, K_BAD, 0, 0);   //  when the block has literals only)
        return what == 1;
    }
    if (left > 0 && !(what != 2 && total == 1 && count[1] == 1)) return false;   // incomplete (a lone 1-bit code is allowed: zlib, Go)
    // symbols sorted by (length, value)
    uint16_t offs[17]; offs[1] = 0;
    for (int l = 1; l <= 15; l++) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    uint16_t sorted[288];
    for (int i = 0; i < nsyms; i++) if (lens[i]) sorted[offs[lens[i]]++] = (uint16_t)i;
    auto entry_of = [&](int sym, int len) -> uint32_t {
        if (what == 0) {
            if (sym < 256) return mk((uint32_t)len, K_LIT, 0, (uint32_t)sym);
            if (sym == 256) return mk((uint32_t)len, K_EOB, 0, 0);
            if (sym > 285) return mk((uint32_t)len, K_BAD, 0, 0);
            return mk((uint32_t)len, K_LEN, LEN_EXTRA[sym - 257], LEN_BASE[sym - 257]);
        }
        if (what == 1) {
            if (sym > 29) return mk((uint32_t)len, K_BAD, 0, 0);
            return mk((uint32_t)len, K_LEN, DIST_EXTRA[sym], DIST_BASE[sym]);
        }
        return mk((uint32_t)len, K_LIT, 0, (uint32_t)sym);
    };
    for (int i = 0; i < prim; i++) table[i] = mk(1, K_BAD, 0, 0);
    // codes are assigned in increasing order; bit-reversed for LSB-first reading
    uint32_t code = 0; int si = 0; int next_sub = prim; uint32_t cur_prefix = ~0u; int cur_sub = 0, cur_sub_bits = 0;
    for (int len = 1; len <= maxlen; len++) {
        for (int c = 0; c < count[len]; c++, si++) {
            const int sym = sorted[si];
            uint32_t rev = 0;
            for (int b = 0; b < len; b++) rev |= ((code >> b) & 1u) << (len - 1 - b);
            if (len <= table_bits) {
                const uint32_t e = entry_of(sym, len);
                for (uint32_t i = rev; i < (uint32_t)prim; i += 1u << len) table[i] = e;
            } else {
                const uint32_t prefix = rev & (uint32_t)(prim - 1);
                if (prefix != cur_prefix) {
                    // size of this subtable: enough bits for the longest code sharing the prefix
                    cur_prefix = prefix;
                    int sub_bits = len - table_bits, rem = 1 << sub_bits;
                    // codes with this prefix are consecutive in canonical order: walk the lengths from here
                    { int l2 = len, cnt = count[len] - c; rem -= cnt;
                      while (rem > 0 && l2 < maxlen) { l2++; sub_bits++; rem = (rem << 1) - count[l2]; } }
                    cur_sub = next_sub; cur_sub_bits = sub_bits;
                    next_sub += 1 << sub_bits;
                    if (next_sub > table_cap) return false;
                    for (int i = cur_sub; i < next_sub; i++) table[i] = mk(1, K_BAD, 0, 0);
                    table[prefix] = mk((uint32_t)table_bits, K_SUB, (uint32_t)sub_bits, (uint32_t)cur_sub);
                }
                const uint32_t e = entry_of(sym, len - table_bits);
                const uint32_t hi = rev >> table_bits;
                for (uint32_t i = hi; i < (1u << cur_sub_bits); i += 1u << (len - table_bits)) table[cur_sub + i] = e;
            }
            code++;
        }
        code <<= 1;
    }
    // Literal pairs (literal/length table only): where a first-level index starts with a literal of l1 bits and its remaining
    // table_bits - l1 bits hold a whole second literal, the entry delivers both — total length in bits 0-3, kind still K_LIT,
    // extra field = 1 | l1 << 1 (bit 8: "two bytes", bits 9-12: the first code's length, for the one-symbol-at-a-time loop),
    // value = lit1 | lit2 << 8.  FASTQ text is mostly literals with 2-6 bit codes: the dependent look-up -> shift -> look-up
    // chain of the fast loop is then walked once per two bytes.  In place, from the top: the second literal's entry sits at
    // i >> l1 < i, not yet rewritten.
    if (what == 0)
        for (int i = prim - 1; i >= 0; i--) {
            const uint32_t e = table[i];
            if ((e & KIND_MASK) != (K_LIT << KIND_SHIFT)) continue;
            const uint32_t l1 = e & 63u;
            if ((int)l1 >= table_bits) continue;
            const uint32_t e2 = table[(uint32_t)i >> l1];
            if ((e2 & (KIND_MASK | 0x1f00u)) != (K_LIT << KIND_SHIFT) || l1 + (e2 & 63u) > (uint32_t)table_bits) continue;
            table[i] = mk(l1 + (e2 & 63u), K_LIT, 1u | (l1 << 1), (e >> 16) | ((e2 >> 16) << 8));
        }
    return true;
}

// both bytes of a literal entry's value are stored, the second counts only for a pair (else it is 0 and overwritten next)
static inline void put2(uint8_t *out, uint32_t e) { const uint16_t v = (uint16_t)(e >> 16); memcpy(out, &v, 2); }

struct Decoder {
    // bit reader
    uint64_t bitbuf = 0; int bitcnt = 0;
    const uint8_t *in = nullptr, *in_end = nullptr;            // current piece of compressed input
    // state across calls
    enum State { HEADER, STORED, CODES, DONE, ERROR } state = HEADER;
    bool last_block = false;
    uint32_t stored_left = 0;
    uint32_t litlen[LITLEN_ENOUGH], dist[DIST_ENOUGH];
    bool have_fixed = false;
    // pending match (a match interrupted because the output piece was full)
    uint32_t pend_len = 0, pend_dist = 0;
    const char *err = nullptr;

    void reset() { bitbuf = 0; bitcnt = 0; state = HEADER; last_block = false; stored_left = 0; pend_len = 0; err = nullptr; }
    void feed(const uint8_t *p, size_t n) { in = p; in_end = p + n; }
    size_t in_left() const { return (size_t)(in_end - in); }
    // bytes of input that are in the bit buffer but not consumed (whole bytes): for handing the tail to the gzip trailer
    void align_to_byte() { const int drop = bitcnt & 7; bitbuf >>= drop; bitcnt -= drop; }
    bool take_byte(uint8_t &b) {
        if (bitcnt >= 8) { b = (uint8_t)bitbuf; bitbuf >>= 8; bitcnt -= 8; return true; }
        if (in < in_end) { b = *in++; return true; }
        return false;
    }

    inline void refill_slow() { while (bitcnt < 56 && in < in_end) { bitbuf |= (uint64_t)(*in++) << bitcnt; bitcnt += 8; } }   // 56..63 bits when input lasts
    bool fail(const char *m) { state = ERROR; err = m; return false; }

    int read_dynamic_header();
    void load_fixed();

    // Decode into [out, out_limit); `hist` = start of the output window (matches may reach back to it, at most 32 KiB
    // are ever needed).  Returns the new output position.  Stops when the output piece is full, the input piece is
    // exhausted (state stays resumable) or the final block ended (state == DONE).  `eof` = no more input will come.
    uint8_t *run(uint8_t *out, uint8_t *out_limit, const uint8_t *hist, bool eof);
    // the one body, compiled twice on x86-64: as it is, and with BMI/BMI2 enabled (shrx/bzhi shorten the shift-and-mask chain
    // between two table look-ups: +10 % on FASTQ text); run() picks by what the CPU it runs on reports
    __attribute__((always_inline)) inline uint8_t *run_body(uint8_t *out, uint8_t *out_limit, const uint8_t *hist, bool eof);
#if defined(__x86_64__)
    __attribute__((target("bmi,bmi2"))) uint8_t *run_bmi2(uint8_t *out, uint8_t *out_limit, const uint8_t *hist, bool eof);
#endif
};

inline void Decoder::load_fixed() {
    uint8_t l[288];
    int i = 0;
    for (; i < 144; i++) l[i] = 8;
    for (; i < 256; i++) l[i] = 9;
    for (; i < 280; i++) l[i] = 7;
    for (; i < 288; i++) l[i] = 8;
    build_table(l, 288, 0, litlen, LITLEN_BITS, LITLEN_ENOUGH);
    uint8_t d[32];
    for (i = 0; i < 32; i++) d[i] = 5;
    build_table(d, 32, 1, dist, DIST_BITS, DIST_ENOUGH);
}

// The header of a dynamic block is at most 14 + 19*3 + 316*(15+7) bits < 900 bytes.  It is parsed from a snapshot of
// the reader, which is committed only when the whole header was there: 1 = done, 0 = more input needed, -1 = error.
inline int Decoder::read_dynamic_header() {
    uint64_t bb = bitbuf; int bc = bitcnt; const uint8_t *ip = in;
    auto top_up = [&]() { while (bc < 56 && ip < in_end) { bb |= (uint64_t)(*ip++) << bc; bc += 8; } };
    auto get = [&](int n) -> uint32_t { const uint32_t v = (uint32_t)(bb & ((1ull << n) - 1)); bb >>= n; bc -= n; return v; };
    top_up();
    if (bc < 14) return 0;
    const int hlit = (int)get(5) + 257, hdist = (int)get(5) + 1, hclen = (int)get(4) + 4;
    if (hlit > 286 || hdist > 30) { fail("flate: corrupt input (too many length or distance symbols)"); return -1; }
    static const uint8_t order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
    uint8_t pl[19] = {0};
    for (int i = 0; i < hclen; i++) { top_up(); if (bc < 3) return 0; pl[order[i]] = (uint8_t)get(3); }
    uint32_t pre[1 << PRE_BITS];
    if (!build_table(pl, 19, 2, pre, PRE_BITS, 1 << PRE_BITS)) { fail("flate: corrupt input (code length codes)"); return -1; }
    uint8_t lens[286 + 30];
    int n = 0; const int total = hlit + hdist;
    while (n < total) {
        top_up();
        const uint32_t e = pre[bb & ((1u << PRE_BITS) - 1)];
        const int cl = (int)(e & 63u);
        if (cl > bc) return 0;
        if (((e >> KIND_SHIFT) & 7u) == K_BAD) { fail("flate: corrupt input (code lengths)"); return -1; }
        const uint32_t sym = e >> 16;
        const int xb = sym == 16 ? 2 : sym == 17 ? 3 : sym == 18 ? 7 : 0;
        if (cl + xb > bc) return 0;
        bb >>= cl; bc -= cl;
        if (sym < 16) { lens[n++] = (uint8_t)sym; continue; }
        int rep; uint8_t v = 0;
        if (sym == 16) { if (n == 0) { fail("flate: corrupt input (repeat without a previous length)"); return -1; } v = lens[n - 1]; rep = 3 + (int)get(2); }
        else if (sym == 17) rep = 3 + (int)get(3);
        else rep = 11 + (int)get(7);
        if (n + rep > total) { fail("flate: corrupt input (repeat past the end)"); return -1; }
        while (rep--) lens[n++] = v;
    }
    if (lens[256] == 0) { fail("flate: corrupt input (no end-of-block code)"); return -1; }
    if (!build_table(lens, hlit, 0, litlen, LITLEN_BITS, LITLEN_ENOUGH)) { fail("flate: corrupt input (literal/length code set)"); return -1; }
    if (!build_table(lens + hlit, hdist, 1, dist, DIST_BITS, DIST_ENOUGH)) { fail("flate: corrupt input (distance code set)"); return -1; }
    bitbuf = bb; bitcnt = bc; in = ip;
    have_fixed = false;
    return 1;
}

#if defined(__x86_64__)
__attribute__((target("bmi,bmi2"))) inline uint8_t *Decoder::run_bmi2(uint8_t *out, uint8_t *out_limit, const uint8_t *hist, bool eof) {
    return run_body(out, out_limit, hist, eof);
}
inline uint8_t *Decoder::run(uint8_t *out, uint8_t *out_limit, const uint8_t *hist, bool eof) {
    static const bool bmi2 = __builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2");
    return bmi2 ? run_bmi2(out, out_limit, hist, eof) : run_body(out, out_limit, hist, eof);
}
#else
inline uint8_t *Decoder::run(uint8_t *out, uint8_t *out_limit, const uint8_t *hist, bool eof) { return run_body(out, out_limit, hist, eof); }
#endif

__attribute__((always_inline)) inline uint8_t *Decoder::run_body(uint8_t *out, uint8_t *out_limit, const uint8_t *hist, bool eof) {
    for (;;) {
        if (state == DONE || state == ERROR) return out;
        if (state == HEADER) {
            refill_slow();
            if (bitcnt < 3) { if (eof) fail("unexpected EOF"); return out; }
            const uint32_t type = (uint32_t)(bitbuf >> 1) & 3u;
            if (type == 0) {
                // stored: LEN / NLEN after the next byte boundary
                uint64_t bb = bitbuf >> 3; int bc = bitcnt - 3;
                const int drop = bc & 7; bb >>= drop; bc -= drop;
                const uint8_t *ip = in;
                while (bc < 32) { if (ip >= in_end) { if (eof) fail("unexpected EOF"); return out; } bb |= (uint64_t)(*ip++) << bc; bc += 8; }
                const uint32_t len = (uint32_t)(bb & 0xffff), nlen = (uint32_t)((bb >> 16) & 0xffff);
                if ((len ^ nlen) != 0xffffu) { fail("flate: corrupt input (stored block lengths)"); return out; }
                last_block = bitbuf & 1u;
                bitbuf = bb >> 32; bitcnt = bc - 32; in = ip;
                stored_left = len; state = STORED;
            } else if (type == 1) {
                last_block = bitbuf & 1u;
                bitbuf >>= 3; bitcnt -= 3;
                if (!have_fixed) { load_fixed(); have_fixed = true; }
                state = CODES;
            } else if (type == 2) {
                const bool lb = bitbuf & 1u;
                const uint64_t sb = bitbuf; const int sc = bitcnt;
                bitbuf >>= 3; bitcnt -= 3;
                const int r = read_dynamic_header();
                if (r < 0) return out;
                if (r == 0) {
                    bitbuf = sb; bitcnt = sc;                       // not all there yet
                    if (eof) fail("unexpected EOF");
                    return out;
                }
                last_block = lb;
                state = CODES;
            } else { fail("flate: corrupt input (block type 3)"); return out; }
            continue;
        }
        if (state == STORED) {
            while (stored_left && bitcnt >= 8 && out < out_limit) { *out++ = (uint8_t)bitbuf; bitbuf >>= 8; bitcnt -= 8; stored_left--; }
            if (stored_left && bitcnt < 8) {
                size_t n = stored_left;
                if (n > in_left()) n = in_left();
                if (n > (size_t)(out_limit - out)) n = (size_t)(out_limit - out);
                memcpy(out, in, n); out += n; in += n; stored_left -= (uint32_t)n;
            }
            if (stored_left) { if (out < out_limit && in_left() == 0 && eof) fail("unexpected EOF"); return out; }
            state = last_block ? DONE : HEADER;
            continue;
        }
        // ---- state == CODES
        if (pend_len) {                                            // finish a match cut by the end of the output piece
            while (pend_len && out < out_limit) { *out = *(out - pend_dist); out++; pend_len--; }
            if (pend_len) return out;
        }
        // fast loop: >= IN_SLACK input bytes and >= OUT_SLACK output bytes ahead.  The reader state lives in locals here:
        // stores through the (byte) output pointer may alias anything, and would force the members to be re-read
        // after every literal.  The loop is software-pipelined: every path ends with "refill, look the NEXT symbol's entry
        // up", and a match does so BEFORE it copies, so that the table load of the next symbol (the head of the dependent
        // chain look-up -> shift -> look-up) is in flight behind the copy.  DNA text inflates as tens of millions of 4-8 byte
        // matches (a 4-letter alphabet repeats every 4-mer within any 32 KiB), not as literals: the per-match latency is
        // what the rate of a FASTQ stream is made of.
        if (in_left() >= IN_SLACK && (size_t)(out_limit - out) >= OUT_SLACK) {
            uint64_t bb = bitbuf; int bc = bitcnt; const uint8_t *ip = in;
            const uint8_t *const ip_stop = in_end - IN_SLACK;
            uint8_t *const out_stop = out_limit - OUT_SLACK;
            const uint32_t *const lt = litlen, *const dt = dist;
            int leave = 0;                                          // 1 = end of block, 2 = error (err set)
            uint64_t w;
#define HULK_REFILL() do { memcpy(&w, ip, 8); bb |= w << bc; ip += (63 - bc) >> 3; bc |= 56; } while (0)
            HULK_REFILL();
            uint32_t e = lt[bb & ((1u << LITLEN_BITS) - 1)];
            // every iteration starts refilled (56..63 bits) with `e` = the entry of the symbol at the reader's position, and makes at
            // most two more refills of <= 7 bytes each: ip <= in_end - IN_SLACK here keeps every 8-byte load inside the piece
            do {
                if (((e >> KIND_SHIFT) & 7u) == K_LIT) {
                    // up to four first-level hits from one refill (each <= 11 bits: one literal or a pair of them)
                    bb >>= (e & 63u); bc -= (int)(e & 63u); put2(out, e); out += 1 + ((e >> 8) & 1u);
                    e = lt[bb & ((1u << LITLEN_BITS) - 1)];
                    if (((e >> KIND_SHIFT) & 7u) != K_LIT) goto not_literal;
                    bb >>= (e & 63u); bc -= (int)(e & 63u); put2(out, e); out += 1 + ((e >> 8) & 1u);
                    e = lt[bb & ((1u << LITLEN_BITS) - 1)];
                    if (((e >> KIND_SHIFT) & 7u) != K_LIT) goto not_literal;
                    bb >>= (e & 63u); bc -= (int)(e & 63u); put2(out, e); out += 1 + ((e >> 8) & 1u);
                    e = lt[bb & ((1u << LITLEN_BITS) - 1)];
                    if (((e >> KIND_SHIFT) & 7u) != K_LIT) goto not_literal;
                    bb >>= (e & 63u); bc -= (int)(e & 63u); put2(out, e); out += 1 + ((e >> 8) & 1u);
                    HULK_REFILL();
                    e = lt[bb & ((1u << LITLEN_BITS) - 1)];
                    continue;
                not_literal:
                    // <= 33 bits used, >= 23 left (the look-up above saw 11 real bits): not enough for a length + distance pair
                    // (<= 48): refill — `e` stays what it is, a refill only adds bits above the ones present
                    HULK_REFILL();
                }
                if (((e >> KIND_SHIFT) & 7u) == K_SUB) { bb >>= LITLEN_BITS; bc -= LITLEN_BITS; e = lt[(e >> 16) + (bb & ((1u << ((e >> 8) & 31u)) - 1))]; }
                const uint64_t sl = bb;                                              // (the length's extra bits are cut out of this)
                bb >>= (e & 63u); bc -= (int)(e & 63u);
                const uint32_t kind = (e >> KIND_SHIFT) & 7u;
                if (kind == K_LIT) {                                                      // (a literal with a long code)
                    *out++ = (uint8_t)(e >> 16);
                    HULK_REFILL();
                    e = lt[bb & ((1u << LITLEN_BITS) - 1)];
                    continue;
                }
                if (kind != K_LEN) {
                    if (kind == K_EOB) { leave = 1; break; }
                    err = "flate: corrupt input (literal/length code)"; leave = 2; break;
                }
                const uint32_t xb = (e >> 8) & 31u;
                const uint32_t len = (e >> 16) + (uint32_t)((sl >> ((e & 63u) - xb)) & ((1u << xb) - 1));
                // <= 15 + 5 bits of >= 56 used so far; the distance needs <= 15 + 13 more
                uint32_t d = dt[bb & ((1u << DIST_BITS) - 1)];
                if (((d >> KIND_SHIFT) & 7u) == K_SUB) { bb >>= DIST_BITS; bc -= DIST_BITS; d = dt[(d >> 16) + (bb & ((1u << ((d >> 8) & 31u)) - 1))]; }
                if (((d >> KIND_SHIFT) & 7u) != K_LEN) { err = "flate: corrupt input (distance code)"; leave = 2; break; }
                const uint64_t sd = bb;
                bb >>= (d & 63u); bc -= (int)(d & 63u);
                const uint32_t dxb = (d >> 8) & 31u;
                const uint32_t distance = (d >> 16) + (uint32_t)((sd >> ((d & 63u) - dxb)) & ((1u << dxb) - 1));
                if (distance > (size_t)(out - hist)) { err = "flate: corrupt input (distance too far back)"; leave = 2; break; }
                // the next symbol's entry, requested before the copy
                HULK_REFILL();
                e = lt[bb & ((1u << LITLEN_BITS) - 1)];
                const uint8_t *src = out - distance;
                uint8_t *dst = out; out += len;
                if (distance >= 8) {
                    // 8 bytes at a time; may write up to 15 bytes past the match (inside OUT_SLACK)
                    uint64_t t;
                    memcpy(&t, src, 8); memcpy(dst, &t, 8); src += 8; dst += 8;
                    memcpy(&t, src, 8); memcpy(dst, &t, 8); src += 8; dst += 8;
                    while (dst < out) { memcpy(&t, src, 8); memcpy(dst, &t, 8); src += 8; dst += 8; }
                } else if (distance == 1) {
                    memset(dst, *src, len);
                } else {
                    while (dst < out) *dst++ = *src++;
                }
            } while (ip <= ip_stop && out <= out_stop);
#undef HULK_REFILL
            // The refill above leaves up to 7 real look-ahead bits of *ip above `bc`; harmless while the next reader ORs the
            // same byte over them, wrong once a stored block has taken its bytes straight from `in` in between: hand the
            // state back with exactly `bc` bits.
            bitbuf = bc < 64 ? bb & ((1ull << bc) - 1) : bb; bitcnt = bc; in = ip;
            if (leave == 1) { state = last_block ? DONE : HEADER; continue; }
            if (leave == 2) { state = ERROR; return out; }
        }
        // careful loop: one symbol at a time with explicit availability checks
        for (;;) {
            if (out >= out_limit) return out;
            refill_slow();
            // an entry can be trusted iff the bits of its code were all there (missing bits read as zeros)
            uint32_t e = litlen[bitbuf & ((1u << LITLEN_BITS) - 1)];
            int used = 0;
            if (((e >> KIND_SHIFT) & 7u) == K_SUB) {
                e = litlen[(e >> 16) + ((bitbuf >> LITLEN_BITS) & ((1u << ((e >> 8) & 31u)) - 1))];
                used = LITLEN_BITS;
            }
            int cl = used + (int)codelen_of(e);
            const uint32_t kind = (e >> KIND_SHIFT) & 7u;
            if (kind == K_LIT && ((e >> 8) & 1u)) cl = (int)((e >> 9) & 15u);      // a pair entry: its first literal only
            if (cl > bitcnt) { if (eof) fail("unexpected EOF"); return out; }
            if (kind == K_BAD) { fail("flate: corrupt input (literal/length code)"); return out; }
            if (kind == K_LIT) { bitbuf >>= cl; bitcnt -= cl; *out++ = (uint8_t)(e >> 16); continue; }
            if (kind == K_EOB) { bitbuf >>= cl; bitcnt -= cl; state = last_block ? DONE : HEADER; goto next_state; }
            // length + distance: needs up to cl + 5 + 15 + 13 = 48+ bits; gather them from a private snapshot
            {
                uint64_t bb = bitbuf; int bc = bitcnt; const uint8_t *ip = in;
                // the bit buffer holds <= 64 bits; consume the length part first, then top up
                const uint32_t xb = (e >> 8) & 31u;
                if (bc < cl + (int)xb) { if (eof) fail("unexpected EOF"); return out; }
                bb >>= cl; bc -= cl;
                const uint32_t len = (e >> 16) + (uint32_t)(bb & ((1u << xb) - 1));
                bb >>= xb; bc -= (int)xb;
                while (bc <= 56 && ip < in_end) { bb |= (uint64_t)(*ip++) << bc; bc += 8; }
                uint32_t d = dist[bb & ((1u << DIST_BITS) - 1)];
                int dused = 0;
                if (((d >> KIND_SHIFT) & 7u) == K_SUB) {
                    if (bc < DIST_BITS) { if (eof) fail("unexpected EOF"); return out; }
                    d = dist[(d >> 16) + ((bb >> DIST_BITS) & ((1u << ((d >> 8) & 31u)) - 1))];
                    dused = DIST_BITS;
                }
                const int dcl = dused + (int)codelen_of(d);
                const uint32_t dk = (d >> KIND_SHIFT) & 7u;
                const uint32_t dxb = (d >> 8) & 31u;
                if (dk != K_LEN) {
                    if (bc >= dcl || eof) { fail(bc >= dcl ? "flate: corrupt input (distance code)" : "unexpected EOF"); }
                    return out;
                }
                if (bc < dcl + (int)dxb) { if (eof) fail("unexpected EOF"); return out; }
                bb >>= dcl; bc -= dcl;
                const uint32_t distance = (d >> 16) + (uint32_t)(bb & ((1u << dxb) - 1));
                bb >>= dxb; bc -= (int)dxb;
                if (distance > (size_t)(out - hist)) { fail("flate: corrupt input (distance too far back)"); return out; }
                bitbuf = bb; bitcnt = bc; in = ip;                   // commit
                uint32_t n = len;
                while (n && out < out_limit) { *out = *(out - distance); out++; n--; }
                if (n) { pend_len = n; pend_dist = distance; return out; }
            }
            // back to the fast loop when there is room again
            if (in_left() >= IN_SLACK && (size_t)(out_limit - out) >= OUT_SLACK) break;
        }
        continue;
    next_state:;
    }
}

}  // namespace inflate
}  // namespace hulk
