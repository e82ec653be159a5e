// Differential test of hulk::inflate (hulk_amd/csrc/fast_inflate.h) against zlib: random inputs of several kinds,
// every compression level and strategy (fixed / dynamic / stored blocks, Huffman-only, RLE; one stream in three mixes
// block kinds: independently compressed chunks joined at Z_FULL_FLUSH points), fed in random input
// pieces (down to 1 byte) with random output pieces, so that every resume point of the decoder is exercised.
// usage: inflate_fuzz [cases] [seed]      exit code 0 = all equal
#include "../../hulk_amd/csrc/fast_inflate.h"
#include "../../hulk_amd/csrc/par_inflate.h"
#include "../../hulk_amd/csrc/crc32_clmul.h"
#include <zlib.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
using namespace hulk::inflate;

static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t> &src, int level, int strategy) {
    z_stream z{};
    deflateInit2(&z, level, Z_DEFLATED, -15, 8, strategy);
    std::vector<uint8_t> out(deflateBound(&z, src.size()) + 64);
    z.next_in = (Bytef *)src.data(); z.avail_in = (uInt)src.size(); z.next_out = out.data(); z.avail_out = (uInt)out.size();
    deflate(&z, Z_FINISH);
    out.resize(z.total_out);
    deflateEnd(&z);
    return out;
}

// One raw stream made of several chunks, each compressed on its own (level, strategy) and closed with Z_FULL_FLUSH —
// a byte-aligned empty stored block, after which no match reaches back — so the chunks concatenate into ONE valid
// stream in which Huffman, fixed and stored (level 0) blocks follow each other in every order; in particular a non-final
// block after a non-empty stored block after a Huffman block.  (deflateParams would do the same inside one z_stream, but
// zlib 1.2.11's level-0 switch reads out of bounds.)
static std::vector<uint8_t> deflate_mixed(const std::vector<uint8_t> &src) {
    std::vector<uint8_t> all;
    size_t pos = 0;
    do {
        const size_t n = std::min(src.size() - pos, (size_t)(rand() % 3 == 0 ? 1 + rand() % 64 : 1 + rand() % 20000));
        const int level = rand() % 3 == 0 ? 0 : rand() % 10;
        const int strat = rand() % 3 == 0 ? Z_FIXED : (rand() % 4 == 0 ? Z_HUFFMAN_ONLY : Z_DEFAULT_STRATEGY);
        const bool last = pos + n >= src.size();
        z_stream z{};
        deflateInit2(&z, level, Z_DEFLATED, -15, 8, strat);
        std::vector<uint8_t> out(deflateBound(&z, n) + 64);
        z.next_in = (Bytef *)src.data() + pos; z.avail_in = (uInt)n; z.next_out = out.data(); z.avail_out = (uInt)out.size();
        deflate(&z, last ? Z_FINISH : Z_FULL_FLUSH);
        all.insert(all.end(), out.begin(), out.begin() + (long)z.total_out);
        deflateEnd(&z);
        pos += n;
    } while (pos < src.size());
    return all;
}

// Hand-made stream: fixed-Huffman blocks of literals and stored blocks in random order, joined at BIT boundaries (no
// flush marker in between) — the shape zlib only produces through deflateParams: a Huffman block whose end-of-block
// code is followed directly by a non-empty stored block and then by another non-final block.
struct BitWriter {
    std::vector<uint8_t> out; uint64_t acc = 0; int n = 0;
    void bits(uint32_t v, int c) { acc |= (uint64_t)v << n; n += c; while (n >= 8) { out.push_back((uint8_t)acc); acc >>= 8; n -= 8; } }
    void huff(uint32_t code, int len) { uint32_t r = 0; for (int i = 0; i < len; i++) r |= ((code >> i) & 1u) << (len - 1 - i); bits(r, len); }
    void align() { if (n) { out.push_back((uint8_t)acc); acc = 0; n = 0; } }
};
static std::vector<uint8_t> deflate_handmade(const std::vector<uint8_t> &src) {
    BitWriter w;
    size_t pos = 0;
    do {
        const size_t n = std::min(src.size() - pos, (size_t)(rand() % 2 ? rand() % 40 : rand() % 30000));
        const bool last = pos + n >= src.size();
        w.bits(last ? 1u : 0u, 1);
        if (rand() % 2) {                                        // stored: LEN, NLEN after the byte boundary
            w.bits(0, 2); w.align();
            w.bits((uint32_t)n, 16); w.bits((uint32_t)n ^ 0xffffu, 16);
            w.out.insert(w.out.end(), src.begin() + (long)pos, src.begin() + (long)(pos + n));
        } else {                                                 // fixed Huffman, literals only (RFC 1951 3.2.6)
            w.bits(1, 2);
            for (size_t i = 0; i < n; i++) { const uint32_t c = src[pos + i]; if (c < 144) w.huff(0x30 + c, 8); else w.huff(0x190 + (c - 144), 9); }
            w.huff(0, 7);
        }
        pos += n;
    } while (pos < src.size());
    w.align();
    return w.out;
}

static bool decode(Decoder &d, const std::vector<uint8_t> &comp, size_t piece, size_t opiece, size_t n_expected,
                   std::vector<uint8_t> &res, std::string &err) {
    d.reset(); d.have_fixed = false;
    std::vector<uint8_t> win(n_expected + 65536 + OUT_SLACK + 1024);
    std::vector<uint8_t> inbuf(comp.size() + 64, 0);
    uint8_t *out = win.data();
    size_t fed = 0;
    d.feed(inbuf.data(), 0);
    for (;;) {
        const size_t tail = d.in_left();
        if (tail) memmove(inbuf.data(), d.in, tail);
        const size_t add = std::min(piece, comp.size() - fed);
        memcpy(inbuf.data() + tail, comp.data() + fed, add); fed += add;
        memset(inbuf.data() + tail + add, 0, 16);
        d.feed(inbuf.data(), tail + add);
        const bool eof = fed == comp.size();
        for (;;) {
            uint8_t *lim = out + opiece;
            uint8_t *end = win.data() + win.size() - OUT_SLACK - 16;
            if (lim > end || lim < out) lim = end;
            out = d.run(out, lim, win.data(), eof);
            if (d.state == Decoder::DONE) { res.assign(win.data(), out); return true; }
            if (d.state == Decoder::ERROR) { err = d.err; return false; }
            if (out < lim) break;                               // starved
            if (lim == end) { err = "output overrun"; return false; }
        }
        if (eof && add == 0) { err = "stuck"; return false; }
    }
}


// par_inflate.h: the symbol decoder on the same streams.  (1) From bit 0 with no window: bytes == the text, no unknown symbol,
// the run stops in front of the final block and `through_final` takes it to the end.  (2) From a block boundary in the middle
// (the boundaries are the ones run (1) passed) with an UNKNOWN window: symbols resolved through the table of the true 32 KiB in
// front == the text from there on.  Any data, not only text: the block search (text only) is not part of this.
static bool spec_check(const std::vector<uint8_t> &src, const std::vector<uint8_t> &comp_in) {
    std::vector<uint8_t> comp(comp_in);
    // (a gzip member has 8 bytes of trailer behind its last block and GzPar lends the decoder IN_SLACK bytes of its zero padding
    // on top for the look-ahead of the last step: the same here for a bare deflate stream)
    comp.resize(comp.size() + 8 + SPEC_IN_SLACK, 0);
    const uint64_t in_bits = 8 * ((uint64_t)comp_in.size() + 8 + IN_SLACK);
    const size_t cap = src.size() + 1024;
    std::vector<uint16_t> sym(SPEC_WINDOW + cap + SPEC_OUT_SLACK + 8);
    std::vector<uint64_t> bounds; std::vector<size_t> outs;
    SpecChunk c;
    c.in = comp.data(); c.in_bits = in_bits; c.base = sym.data() + SPEC_WINDOW; c.cap = cap; c.hist_have = 0;
    spec_run(c, 0, [&](uint64_t pos) { bounds.push_back(pos); outs.push_back((size_t)0); return false; }, true);
    if (c.stop != SPEC_LINK || c.out_len != src.size()) { printf("spec: stop %d out %zu of %zu\n", (int)c.stop, c.out_len, src.size()); return false; }
    for (size_t i = 0; i < src.size(); i++) if (c.base[i] != src[i]) { printf("spec: byte %zu differs\n", i); return false; }
    // where each boundary is in the text: decode again, stopping there
    if (bounds.size() < 2) return true;
    const size_t pick = 1 + (size_t)rand() % (bounds.size() - 1);
    SpecChunk a;
    a.in = comp.data(); a.in_bits = in_bits; a.base = sym.data() + SPEC_WINDOW; a.cap = cap; a.hist_have = 0;
    spec_run(a, 0, [&](uint64_t pos) { return pos >= bounds[pick]; }, false);
    if (a.stop != SPEC_LINK || a.end_bit != bounds[pick]) { printf("spec: no stop at boundary %zu (stop %d)\n", pick, (int)a.stop); return false; }
    const size_t at = a.out_len;
    // the true window in front of `at` (right-aligned), as a table
    std::vector<uint8_t> lut(256 + SPEC_WINDOW, 0);
    for (int i = 0; i < 256; i++) lut[i] = (uint8_t)i;
    for (size_t i = 0; i < SPEC_WINDOW; i++) if (at + i >= SPEC_WINDOW) lut[256 + i] = src[at + i - SPEC_WINDOW];
    for (size_t i = 0; i < SPEC_WINDOW; i++) sym[i] = (uint16_t)(256 + i);
    SpecChunk b;
    b.in = comp.data(); b.in_bits = in_bits; b.base = sym.data() + SPEC_WINDOW; b.cap = cap; b.hist_have = SPEC_WINDOW;
    spec_run(b, bounds[pick], [](uint64_t) { return false; }, true);
    if (b.stop != SPEC_LINK || at + b.out_len != src.size()) { printf("spec: from boundary %zu: stop %d, %zu + %zu of %zu\n", pick, (int)b.stop, at, b.out_len, src.size()); return false; }
    std::vector<uint8_t> res(b.out_len);
    spec_resolve(b.base, b.out_len, lut.data(), res.data());
    if (memcmp(res.data(), src.data() + at, b.out_len) != 0) { printf("spec: resolved text differs from boundary %zu on\n", pick); return false; }
    return true;
}

// crc32_clmul.h against zlib: random lengths, alignments, chained calls
static bool crc_check() {
    std::vector<uint8_t> buf(70000);
    for (auto &x : buf) x = (uint8_t)rand();
    for (int it = 0; it < 3000; it++) {
        const size_t off = (size_t)rand() % 64, n = rand() % 4 == 0 ? (size_t)rand() % 200 : (size_t)rand() % (buf.size() - 64);
        const uint32_t start = it % 2 ? (uint32_t)rand() * 2654435761u : 0;
        if (hulk::crc32_fast(start, buf.data() + off, n) != (uint32_t)crc32(start, buf.data() + off, (uInt)n)) { printf("crc32_fast differs: off %zu n %zu\n", off, n); return false; }
    }
    return true;
}

int main(int argc, char **argv) {
    const int cases = argc > 1 ? atoi(argv[1]) : 400;
    srand(argc > 2 ? atoi(argv[2]) : 1);
    Decoder *d = new Decoder;
    int bad = 0;
    for (int it = 0; it < cases; it++) {
        const size_t n = rand() % 4 == 0 ? rand() % 2000 : rand() % 300000;
        std::vector<uint8_t> src(n);
        const int kind = rand() % 5;
        for (size_t i = 0; i < n; i++) {
            if (kind == 0) src[i] = (uint8_t)rand();
            else if (kind == 1) src[i] = "ACGT"[rand() & 3];
            else if (kind == 2) src[i] = (i % 150 < 100) ? "ACGTN"[rand() % 5] : 'I';
            else if (kind == 3) src[i] = (uint8_t)(i * 7 / 13);
            else src[i] = (rand() % 50) ? 'A' : (uint8_t)rand();
        }
        const int level = rand() % 10;
        const int strat = (rand() % 4 == 0) ? Z_FIXED : (rand() % 5 == 0 ? Z_HUFFMAN_ONLY : (rand() % 7 == 0 ? Z_RLE : Z_DEFAULT_STRATEGY));
        const bool mixed = it % 3 == 2;                      // every third case: blocks of different kinds in one stream
        const std::vector<uint8_t> comp = mixed ? (it % 2 ? deflate_handmade(src) : deflate_mixed(src)) : deflate_raw(src, level, strat);
        const size_t piece = rand() % 3 == 0 ? 1 + rand() % 40 : (rand() % 2 ? (size_t)1 << 20 : 1 + rand() % 5000);
        const size_t opiece = rand() % 3 == 0 ? 1 + rand() % 600 : (size_t)1 << 22;
        std::vector<uint8_t> res; std::string err;
        const bool ok = decode(*d, comp, piece, opiece, n, res, err);
        if (!spec_check(src, comp)) { bad++; printf("FAIL spec it=%d mixed=%d n=%zu kind=%d level=%d strat=%d\n", it, (int)mixed, n, kind, level, strat); }
        if (!ok || res != src) {
            bad++;
            if (bad < 10) printf("FAIL it=%d mixed=%d n=%zu kind=%d level=%d strat=%d piece=%zu opiece=%zu ok=%d err=%s got=%zu\n",
                                 it, (int)mixed, n, kind, level, strat, piece, opiece, (int)ok, err.c_str(), res.size());
        }
        // a truncated stream must end in an error, never in DONE with wrong data
        if (comp.size() > 8 && it % 7 == 0) {
            std::vector<uint8_t> cut(comp.begin(), comp.begin() + (long)(comp.size() - 1 - (size_t)rand() % std::min<size_t>(comp.size() - 1, 50)));
            std::vector<uint8_t> r2; std::string e2;
            if (decode(*d, cut, piece, opiece, n, r2, e2) && r2 != src) { bad++; printf("FAIL truncated it=%d accepted\n", it); }
        }
    }
    // mangled streams: any outcome but a crash or an out-of-bounds access is fine (build with -fsanitize=address to see those)
    int rejected = 0;
    for (int it = 0; it < cases; it++) {
        const size_t n = 1 + rand() % 50000;
        std::vector<uint8_t> src(n);
        for (size_t i = 0; i < n; i++) src[i] = (i % 97 < 60) ? "ACGT"[rand() & 3] : (uint8_t)('A' + rand() % 40);
        std::vector<uint8_t> comp = deflate_raw(src, rand() % 10, rand() % 3 ? Z_DEFAULT_STRATEGY : Z_FIXED);
        for (int f = 1 + rand() % 4; f > 0; f--) comp[(size_t)rand() % comp.size()] ^= (uint8_t)(1u << (rand() % 8));
        if (rand() % 5 == 0) comp.resize(1 + (size_t)rand() % comp.size());
        std::vector<uint8_t> res; std::string err;
        // the output window is sized for 4x the original: a mangled stream may inflate to more than that
        if (!decode(*d, comp, rand() % 2 ? 1 + rand() % 300 : (size_t)1 << 20, (size_t)1 << 22, 4 * n + 70000, res, err)) rejected++;
        {   // the symbol decoder on the same mangled stream, from bit 0 and from an arbitrary bit with an unknown window
            std::vector<uint8_t> padded(comp); padded.resize(comp.size() + SPEC_IN_SLACK, 0);
            const size_t cap = 4 * n + 70000;
            std::vector<uint16_t> sym(SPEC_WINDOW + cap + SPEC_OUT_SLACK + 8);
            for (size_t i = 0; i < SPEC_WINDOW; i++) sym[i] = (uint16_t)(256 + i);
            SpecChunk c;
            c.in = padded.data(); c.in_bits = 8 * (uint64_t)comp.size(); c.base = sym.data() + SPEC_WINDOW; c.cap = cap; c.hist_have = 0;
            spec_run(c, 0, [](uint64_t) { return false; }, true);
            c.hist_have = SPEC_WINDOW;
            spec_run(c, (uint64_t)rand() % (8 * comp.size()), [](uint64_t) { return false; }, rand() % 2);
            (void)find_block_start(padded.data(), c.in_bits, 0, c.in_bits);
        }
    }
    if (!crc_check()) bad++;
    printf("%d cases, %d bad; %d mangled streams, %d rejected; crc32_fast %s\n", cases, bad, cases, rejected, hulk::crc32_fast_usable() ? "folds (PCLMULQDQ)" : "is zlib's");
    delete d;
    return bad != 0;
}
