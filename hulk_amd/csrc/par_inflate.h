// par_inflate.h — decoding the MIDDLE of a DEFLATE stream, so that one gzip member can be inflated by several threads.
//
// A `.gz` written by gzip / pigz / a sequencer's software is ONE member: a chain of blocks whose back-references reach up to
// 32 KiB into text that only exists once everything before has been inflated — one thread, 1.2-1.5 GB/s of FASTQ text, which
// is what bounds a `.gz` run end to end (DESIGN §3 "Host ingest").  The way around it (the idea is pugz's, Kerbiriou &
// Chikhi 2019; nothing of its code is used): a thread that starts at a block boundary somewhere in the file can decode
// everything EXCEPT the bytes that copy from the 32 KiB in front of its first byte — so it decodes into 16-bit symbols,
// 0..255 a known byte, 256 + i "byte i of the window in front of me", matches copy symbols like bytes, and a second pass
// replaces the unknown ones through a 33,024-entry table once the thread in front has delivered its last 32 KiB.
//
// Where a block starts is not marked in the stream.  `find_block_start` tries bit positions in turn for the header of a
// non-final dynamic block whose three code sets are complete and whose literal codes are text bytes only (FASTQ / FASTA is
// text: a block of anything else is simply never found, and the caller falls back to the one-thread reader).  A position that
// passes is only a CANDIDATE: the caller accepts a chunk's output only if the decoder of the chunk in front of it arrived,
// at a block boundary, at exactly that bit (hulk_ingest.hip, GzPar) — a false candidate costs time, never correctness.
//
// A decoder stops in front of the final block (SPEC_FINAL).  The caller may then decode that one block with the window it
// knows by then (`through_final`) and check the trailer itself, to go on with a member that follows; whenever anything about
// that fails, and at the last member's end, the one-thread reader takes the stream over from the bit in front of the final
// block with the window, CRC-32 and length so far — so every error message, the handling of trailing bytes and of a cut file
// stay the one-thread reader's.
#pragma once
#include "fast_inflate.h"

namespace hulk {
namespace inflate {

constexpr size_t SPEC_WINDOW = 32768;
constexpr size_t SPEC_OUT_SLACK = 320;      // symbols a decoder may write past `cap`
constexpr size_t SPEC_IN_SLACK = 32;        // readable bytes the caller guarantees past the valid input (any value)

enum SpecStop {
    SPEC_LINK = 0,      // the caller's stop rule said so, at a block boundary
    SPEC_FINAL,         // the next block is the stream's last one
    SPEC_ERROR,         // the block after the boundary does not decode
    SPEC_CAP,           // ... does not fit the symbol buffer
    SPEC_INPUT          // ... is not completely in the input
};

static inline uint64_t peek_bits(const uint8_t *in, uint64_t pos) {      // >= 57 bits from bit `pos` on
    uint64_t w; memcpy(&w, in + (pos >> 3), 8); return w >> (pos & 7);
}
static inline bool is_text_byte(int c) { return c == 9 || c == 10 || c == 13 || (c >= 32 && c <= 126); }

struct SpecTables { uint32_t litlen[LITLEN_ENOUGH], dist[DIST_ENOUGH]; };

// Header of a dynamic block (the bits after BFINAL/BTYPE) at bit `pos`.  Returns the bit after it; 0 = not a valid header
// (or, with text_only, one with a code for a byte that is not text); 1 = the input ends inside it.
static inline uint64_t parse_dynamic_at(const uint8_t *in, uint64_t in_bits, uint64_t pos, SpecTables &t, bool text_only) {
    if (pos + 14 > in_bits) return 1;
    uint64_t bb = peek_bits(in, pos);
    const int hlit = (int)(bb & 31) + 257, hdist = (int)((bb >> 5) & 31) + 1, hclen = (int)((bb >> 10) & 15) + 4;
    if (hlit > 286 || hdist > 30) return 0;
    pos += 14;
    if (pos + 3 * (uint64_t)hclen > in_bits) return 1;
    bb = peek_bits(in, pos);                                               // 19 * 3 = 57 bits
    static const uint8_t order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
    uint8_t pl[19] = {0};
    int kraft = 0;
    for (int i = 0; i < hclen; i++) { const int l = (int)((bb >> (3 * i)) & 7); pl[order[i]] = (uint8_t)l; if (l) kraft += 128 >> l; }
    if (kraft != 128) return 0;                                            // (build_table would say the same, later)
    pos += 3 * (uint64_t)hclen;
    uint32_t pre[1 << PRE_BITS];
    if (!build_table(pl, 19, 2, pre, PRE_BITS, 1 << PRE_BITS)) return 0;
    uint8_t lens[286 + 30];
    int n = 0; const int total = hlit + hdist;
    while (n < total) {
        if (pos + 14 > in_bits) return 1;                                  // (a code of <= 7 bits and <= 7 extra bits)
        bb = peek_bits(in, pos);
        const uint32_t e = pre[bb & ((1u << PRE_BITS) - 1)];
        if (((e >> KIND_SHIFT) & 7u) == K_BAD) return 0;
        const int cl = (int)(e & 63u);
        const uint32_t sym = e >> 16;
        bb >>= cl; pos += (uint64_t)cl;
        if (sym < 16) { lens[n++] = (uint8_t)sym; continue; }
        int rep; uint8_t v = 0;
        if (sym == 16) { if (n == 0) return 0; v = lens[n - 1]; rep = 3 + (int)(bb & 3); pos += 2; }
        else if (sym == 17) { rep = 3 + (int)(bb & 7); pos += 3; }
        else { rep = 11 + (int)(bb & 127); pos += 7; }
        if (n + rep > total) return 0;
        while (rep--) lens[n++] = v;
    }
    if (lens[256] == 0) return 0;
    if (text_only) for (int c = 0; c < 256; c++) if (lens[c] && !is_text_byte(c)) return 0;
    if (!build_table(lens, hlit, 0, t.litlen, LITLEN_BITS, LITLEN_ENOUGH)) return 0;
    if (!build_table(lens + hlit, hdist, 1, t.dist, DIST_BITS, DIST_ENOUGH)) return 0;
    return pos;
}

static inline const SpecTables &spec_fixed_tables() {
    static const SpecTables *const fx = [] {
        SpecTables *t = new SpecTables;
        uint8_t l[288]; int i = 0;
        for (; i < 144; i++) l[i] = 8;
        for (; i < 256; i++) l[i] = 9;
        for (; i < 280; i++) l[i] = 7;
        for (; i < 288; i++) l[i] = 8;
        build_table(l, 288, 0, t->litlen, LITLEN_BITS, LITLEN_ENOUGH);
        uint8_t d[32];
        for (i = 0; i < 32; i++) d[i] = 5;
        build_table(d, 32, 1, t->dist, DIST_BITS, DIST_ENOUGH);
        return t;
    }();
    return *fx;
}

// a literal entry's one or two bytes as symbols (the second counts only for a pair: else it is 0 and overwritten next)
static inline void put2w(uint16_t *out, uint32_t e) { const uint32_t v = ((e >> 16) & 0xffu) | ((e >> 24) << 16); memcpy(out, &v, 4); }

// The symbols of one block, from bit `pos` (behind its header) to its end-of-block code.  0 = done (pos, out advanced),
// SPEC_ERROR / SPEC_CAP / SPEC_INPUT otherwise (pos and out are then meaningless).  `hist_lo` = the lowest symbol a match
// may copy from.  The loop is the one of Decoder::run_body (fast_inflate.h) on 16-bit output; the whole input is in memory,
// so nothing here is resumable.
__attribute__((always_inline)) static inline int spec_block_body(const uint8_t *in, const uint8_t *ip_stop, uint64_t &pos, uint16_t *&outp, uint16_t *out_stop,
                                                                 const uint16_t *hist_lo, const uint32_t *lt, const uint32_t *dt) {
    const uint8_t *ip = in + (pos >> 3);
    if (ip > ip_stop) return SPEC_INPUT;
    uint16_t *out = outp;
    uint64_t bb = 0, w; int bc = 0;
#define HULK_REFILL() do { memcpy(&w, ip, 8); bb |= w << bc; ip += (63 - bc) >> 3; bc |= 56; } while (0)
    HULK_REFILL();
    bb >>= (pos & 7); bc -= (int)(pos & 7);
    HULK_REFILL();
    uint32_t e = lt[bb & ((1u << LITLEN_BITS) - 1)];
    int ret;
    for (;;) {
        // every iteration starts refilled (56..63 bits) with `e` = the entry of the symbol at the reader's position and makes at most
        // two more refills: ip <= in_end - IN_SLACK keeps every 8-byte load inside the input
        if (ip > ip_stop) { ret = SPEC_INPUT; break; }
        if (out > out_stop) { ret = SPEC_CAP; break; }
        if (((e >> KIND_SHIFT) & 7u) == K_LIT) {
            bb >>= (e & 63u); bc -= (int)(e & 63u); put2w(out, e); out += 1 + ((e >> 8) & 1u);
            e = lt[bb & ((1u << LITLEN_BITS) - 1)];
            if (((e >> KIND_SHIFT) & 7u) != K_LIT) goto not_literal;
            bb >>= (e & 63u); bc -= (int)(e & 63u); put2w(out, e); out += 1 + ((e >> 8) & 1u);
            e = lt[bb & ((1u << LITLEN_BITS) - 1)];
            if (((e >> KIND_SHIFT) & 7u) != K_LIT) goto not_literal;
            bb >>= (e & 63u); bc -= (int)(e & 63u); put2w(out, e); out += 1 + ((e >> 8) & 1u);
            e = lt[bb & ((1u << LITLEN_BITS) - 1)];
            if (((e >> KIND_SHIFT) & 7u) != K_LIT) goto not_literal;
            bb >>= (e & 63u); bc -= (int)(e & 63u); put2w(out, e); out += 1 + ((e >> 8) & 1u);
            HULK_REFILL();
            e = lt[bb & ((1u << LITLEN_BITS) - 1)];
            continue;
        not_literal:
            HULK_REFILL();
        }
        if (((e >> KIND_SHIFT) & 7u) == K_SUB) { bb >>= LITLEN_BITS; bc -= LITLEN_BITS; e = lt[(e >> 16) + (bb & ((1u << ((e >> 8) & 31u)) - 1))]; }
        const uint64_t sl = bb;
        bb >>= (e & 63u); bc -= (int)(e & 63u);
        const uint32_t kind = (e >> KIND_SHIFT) & 7u;
        if (kind == K_LIT) {
            *out++ = (uint16_t)((e >> 16) & 0xffu);
            HULK_REFILL();
            e = lt[bb & ((1u << LITLEN_BITS) - 1)];
            continue;
        }
        if (kind != K_LEN) { ret = kind == K_EOB ? 0 : SPEC_ERROR; break; }
        const uint32_t xb = (e >> 8) & 31u;
        const uint32_t len = (e >> 16) + (uint32_t)((sl >> ((e & 63u) - xb)) & ((1u << xb) - 1));
        uint32_t d = dt[bb & ((1u << DIST_BITS) - 1)];
        if (((d >> KIND_SHIFT) & 7u) == K_SUB) { bb >>= DIST_BITS; bc -= DIST_BITS; d = dt[(d >> 16) + (bb & ((1u << ((d >> 8) & 31u)) - 1))]; }
        if (((d >> KIND_SHIFT) & 7u) != K_LEN) { ret = SPEC_ERROR; break; }
        const uint64_t sd = bb;
        bb >>= (d & 63u); bc -= (int)(d & 63u);
        const uint32_t dxb = (d >> 8) & 31u;
        const uint32_t distance = (d >> 16) + (uint32_t)((sd >> ((d & 63u) - dxb)) & ((1u << dxb) - 1));
        if (distance > (size_t)(out - hist_lo)) { ret = SPEC_ERROR; break; }
        HULK_REFILL();
        e = lt[bb & ((1u << LITLEN_BITS) - 1)];
        const uint16_t *src = out - distance;
        uint16_t *dst = out; out += len;
        if (distance >= 4) {
            // four symbols at a time; may write up to 7 symbols past the match (inside SPEC_OUT_SLACK)
            uint64_t t;
            memcpy(&t, src, 8); memcpy(dst, &t, 8); src += 4; dst += 4;
            while (dst < out) { memcpy(&t, src, 8); memcpy(dst, &t, 8); src += 4; dst += 4; }
        } else {
            while (dst < out) *dst++ = *src++;
        }
    }
#undef HULK_REFILL
    if (ret != 0) return ret;
    pos = 8 * (uint64_t)(ip - in) - (uint64_t)bc;
    outp = out;
    return 0;
}
#if defined(__x86_64__)
__attribute__((target("bmi,bmi2"))) static inline int spec_block_bmi2(const uint8_t *in, const uint8_t *ip_stop, uint64_t &pos, uint16_t *&out, uint16_t *out_stop,
                                                                        const uint16_t *hist_lo, const uint32_t *lt, const uint32_t *dt) {
    return spec_block_body(in, ip_stop, pos, out, out_stop, hist_lo, lt, dt);
}
#endif
static inline int spec_block(const uint8_t *in, const uint8_t *ip_stop, uint64_t &pos, uint16_t *&out, uint16_t *out_stop,
                             const uint16_t *hist_lo, const uint32_t *lt, const uint32_t *dt) {
#if defined(__x86_64__)
    static const bool bmi2 = __builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2");
    if (bmi2) return spec_block_bmi2(in, ip_stop, pos, out, out_stop, hist_lo, lt, dt);
#endif
    return spec_block_body(in, ip_stop, pos, out, out_stop, hist_lo, lt, dt);
}

struct SpecChunk {
    // input: `in_bits` valid bits at `in` (SPEC_IN_SLACK readable bytes behind them); symbols go to base[0 .. cap), and
    // base[-SPEC_WINDOW .. 0) is the window the caller filled: 256 + i for an unknown one, the bytes themselves for a known one
    // (of which the last `hist_have` exist: a match may not reach further back)
    const uint8_t *in = nullptr; uint64_t in_bits = 0;
    uint16_t *base = nullptr; size_t cap = 0; size_t hist_have = SPEC_WINDOW;
    // result: the last block boundary reached and the symbols in front of it; why the decoder went no further
    uint64_t end_bit = 0; size_t out_len = 0; SpecStop stop = SPEC_ERROR; uint64_t blocks = 0;
};

// Decode block after block from the boundary at `start_bit`; `should_stop(bit)` is asked at every boundary (the first included).
// `through_final`: the final block is decoded too and ends the run (SPEC_LINK, end_bit = the bit behind its end-of-block code).
template <class StopFn>
static inline void spec_run(SpecChunk &c, uint64_t start_bit, StopFn &&should_stop, bool through_final = false) {
    SpecTables tb;
    uint64_t pos = start_bit;
    uint16_t *out = c.base;
    const uint64_t in_bytes = c.in_bits >> 3;
    const uint8_t *const ip_stop = in_bytes >= IN_SLACK ? c.in + in_bytes - IN_SLACK : c.in - 1;
    uint16_t *const out_stop = c.base + c.cap - 258;
    const uint16_t *const hist_lo = c.base - c.hist_have;
    c.blocks = 0;
    for (;;) {
        c.end_bit = pos; c.out_len = (size_t)(out - c.base);
        if (should_stop(pos)) { c.stop = SPEC_LINK; return; }
        if (pos + 3 > c.in_bits) { c.stop = SPEC_INPUT; return; }
        const uint64_t hb = peek_bits(c.in, pos);
        const bool is_final = hb & 1u;
        if (is_final && !through_final) { c.stop = SPEC_FINAL; return; }
        const uint32_t type = (uint32_t)(hb >> 1) & 3u;
        pos += 3;
        if (type == 0) {
            pos = (pos + 7) & ~(uint64_t)7;
            if (pos + 32 > c.in_bits) { c.stop = SPEC_INPUT; return; }
            const uint8_t *p = c.in + (pos >> 3);
            const uint32_t len = (uint32_t)p[0] | ((uint32_t)p[1] << 8), nlen = (uint32_t)p[2] | ((uint32_t)p[3] << 8);
            if ((len ^ nlen) != 0xffffu) { c.stop = SPEC_ERROR; return; }
            pos += 32;
            if (pos + 8 * (uint64_t)len > c.in_bits) { c.stop = SPEC_INPUT; return; }
            if ((size_t)(out - c.base) + len > c.cap) { c.stop = SPEC_CAP; return; }
            p += 4;
            for (uint32_t i = 0; i < len; i++) out[i] = p[i];
            out += len; pos += 8 * (uint64_t)len;
            c.blocks++;
            if (is_final) { c.end_bit = pos; c.out_len = (size_t)(out - c.base); c.stop = SPEC_LINK; return; }
            continue;
        }
        if (type == 3) { c.stop = SPEC_ERROR; return; }
        const SpecTables *t = &tb;
        if (type == 1) t = &spec_fixed_tables();
        else {
            const uint64_t np = parse_dynamic_at(c.in, c.in_bits, pos, tb, false);
            if (np < 2) { c.stop = np ? SPEC_INPUT : SPEC_ERROR; return; }
            pos = np;
        }
        if (out > out_stop) { c.stop = SPEC_CAP; return; }
        const int r = spec_block(c.in, ip_stop, pos, out, out_stop, hist_lo, t->litlen, t->dist);
        if (r != 0) { c.stop = (SpecStop)r; return; }
        c.blocks++;
        if (is_final) { c.end_bit = pos; c.out_len = (size_t)(out - c.base); c.stop = SPEC_LINK; return; }
    }
}

// First bit in [from, to) that starts a plausible non-final dynamic block of text (see the head of this file); ~0 = none.
static inline uint64_t find_block_start(const uint8_t *in, uint64_t in_bits, uint64_t from, uint64_t to) {
    SpecTables tbs, *tb = &tbs;                 // (on the stack: this runs on worker threads, where a failed `new` would end the process)
    uint64_t found = ~0ull;
    if (to > in_bits) to = in_bits;
    for (uint64_t p = from; p + 17 <= to; p++) {
        const uint64_t bb = peek_bits(in, p);
        if ((bb & 7u) != 4u) continue;                                     // BFINAL 0, BTYPE 2 (bits 1-2 = 0, 1)
        if (((bb >> 3) & 31u) > 29u || ((bb >> 8) & 31u) > 29u) continue;  // HLIT, HDIST
        const uint64_t np = parse_dynamic_at(in, in_bits, p + 3, *tb, true);
        if (np < 2) continue;
        found = p;
        break;
    }
    return found;
}

// symbols -> bytes through lut[0..255] = identity, lut[256 + i] = byte i of the window (one table look-up per symbol, known or
// not: a test for "all eight known" in front of it mispredicts on DNA text and was slower: 1.0-1.4 against 1.5 GB/s)
static inline void spec_resolve(const uint16_t *sym, size_t n, const uint8_t *lut, uint8_t *dst) {
    for (size_t i = 0; i < n; i++) dst[i] = lut[sym[i]];
}

}  // namespace inflate
}  // namespace hulk
