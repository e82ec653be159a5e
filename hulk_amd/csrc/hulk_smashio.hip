// hulk_smashio.hip — the directory form of `hulk smash` in native code (host side; the N x N x S comparison itself is k_smash,
// hulk_cws.hip): LoadHULKdata for every sketch file on a pool of threads (src/sketchio/sketchio.go:100-195: parse the JSON,
// class / version checks, MD5 of the little-endian mins against the stored md5sum, src/helpers/helpers.go:156-166),
// FindSketch per file (sketchio.go:198-257), the length check of GetDistance (sketchio.go:274-277), the matrix on the GPU and
// the CSV encoding/csv would write (cmd/smash.go:183-226).  Error texts are the reference's.  Until round 6 this was a serial
// Python loop (json + hashlib: 0.47 of the 0.58 s a 1024-file run took, 300x the GPU time it fed).
#include "hulk_ctx.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <thread>

namespace hulk {
namespace {

// ---- MD5 (RFC 1321) -------------------------------------------------------------------------------------------------
struct Md5 {
    uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
    static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
    void block(const uint8_t *p) {
        static const uint32_t K[64] = {
            0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1,
            0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453,
            0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942,
            0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05,
            0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d,
            0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
        static const int R[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
                                  4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
        uint32_t w[16];
        memcpy(w, p, 64);                                        // (little-endian host)
        uint32_t A = a, B = b, C = c, D = d;
        for (int i = 0; i < 64; i++) {
            uint32_t f; int g;
            if (i < 16) { f = (B & C) | (~B & D); g = i; }
            else if (i < 32) { f = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
            else if (i < 48) { f = B ^ C ^ D; g = (3 * i + 5) & 15; }
            else { f = C ^ (B | ~D); g = (7 * i) & 15; }
            const uint32_t t = D; D = C; C = B;
            B = B + rol(A + f + K[i] + w[g], R[i]);
            A = t;
        }
        a += A; b += B; c += C; d += D;
    }
    // hex digest of `n` bytes
    static std::string hex(const uint8_t *p, size_t n) {
        Md5 m;
        size_t i = 0;
        for (; i + 64 <= n; i += 64) m.block(p + i);
        uint8_t tail[128] = {0};
        const size_t rem = n - i;
        memcpy(tail, p + i, rem);
        tail[rem] = 0x80;
        const size_t padded = rem + 1 + 8 <= 64 ? 64 : 128;
        const uint64_t bits = (uint64_t)n * 8;
        memcpy(tail + padded - 8, &bits, 8);
        m.block(tail);
        if (padded == 128) m.block(tail + 64);
        uint32_t out[4] = {m.a, m.b, m.c, m.d};
        static const char *hx = "0123456789abcdef";
        std::string s(32, '0');
        const uint8_t *o = (const uint8_t *)out;
        for (int k = 0; k < 16; k++) { s[2 * k] = hx[o[k] >> 4]; s[2 * k + 1] = hx[o[k] & 15]; }
        return s;
    }
};

// ---- a small JSON reader: enough of RFC 8259 for what encoding/json accepts in these files ----------------------------------
struct Sig {
    std::string algo, md5;
    bool has_algo = false, has_sketch = false;
    uint64_t ksize = 0;
    std::vector<uint64_t> mins;
    std::vector<double> weights;
};
struct Doc {
    std::string cls, filename, hashfn, license, version, banner;
    bool has[6] = {false, false, false, false, false, false};     // class, filename, hash_function, license, version, banner_label (as strings)
    bool has_sigs = false;
    std::vector<Sig> sigs;
};

struct Reader {
    const char *p, *end;
    std::string err;
    int depth = 0;                                                // nesting of skipped values (encoding/json gives up at 10000 too)
    bool fail(const char *what) { if (err.empty()) err = what; return false; }
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
    bool lit(const char *s) { const size_t n = strlen(s); if ((size_t)(end - p) >= n && memcmp(p, s, n) == 0) { p += n; return true; } return false; }
    static void utf8(std::string &o, uint32_t cp) {
        if (cp < 0x80) o += (char)cp;
        else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 63)); }
        else if (cp < 0x10000) { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 63)); o += (char)(0x80 | (cp & 63)); }
        else { o += (char)(0xF0 | (cp >> 18)); o += (char)(0x80 | ((cp >> 12) & 63)); o += (char)(0x80 | ((cp >> 6) & 63)); o += (char)(0x80 | (cp & 63)); }
    }
    bool hex4(uint32_t &v) {
        if (end - p < 4) return fail("bad \\u escape");
        v = 0;
        for (int i = 0; i < 4; i++) {
            const char c = *p++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else return fail("bad \\u escape");
        }
        return true;
    }
    bool str(std::string *out) {                                  // at '"'
        if (p >= end || *p != '"') return fail("expected a string");
        p++;
        for (;;) {
            if (p >= end) return fail("unterminated string");
            const unsigned char c = (unsigned char)*p++;
            if (c == '"') return true;
            if (c < 0x20) return fail("control character in string");
            if (c != '\\') { if (out) *out += (char)c; continue; }
            if (p >= end) return fail("unterminated string");
            const char e = *p++;
            char lit_c = 0;
            switch (e) {
                case '"': lit_c = '"'; break; case '\\': lit_c = '\\'; break; case '/': lit_c = '/'; break;
                case 'b': lit_c = '\b'; break; case 'f': lit_c = '\f'; break; case 'n': lit_c = '\n'; break;
                case 'r': lit_c = '\r'; break; case 't': lit_c = '\t'; break;
                case 'u': {
                    uint32_t cp;
                    if (!hex4(cp)) return false;
                    if (cp >= 0xD800 && cp < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {      // surrogate pair
                        const char *save = p; p += 2;
                        uint32_t lo;
                        if (!hex4(lo)) return false;
                        if (lo >= 0xDC00 && lo < 0xE000) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        else { p = save; cp = 0xFFFD; }
                    } else if (cp >= 0xD800 && cp < 0xE000) cp = 0xFFFD;     // encoding/json: unpaired surrogate -> U+FFFD
                    if (out) utf8(*out, cp);
                    continue;
                }
                default: return fail("bad escape in string");
            }
            if (out) *out += lit_c;
        }
    }
    // a JSON number token [b, e); grammar checked here, the value by the caller
    bool num(const char *&b, const char *&e, bool &integral) {
        b = p; integral = true;
        if (p < end && *p == '-') p++;
        if (p >= end) return fail("bad number");
        if (*p == '0') p++;
        else if (*p >= '1' && *p <= '9') { while (p < end && *p >= '0' && *p <= '9') p++; }
        else return fail("bad number");
        if (p < end && *p == '.') { integral = false; p++; if (p >= end || *p < '0' || *p > '9') return fail("bad number"); while (p < end && *p >= '0' && *p <= '9') p++; }
        if (p < end && (*p == 'e' || *p == 'E')) {
            integral = false; p++;
            if (p < end && (*p == '+' || *p == '-')) p++;
            if (p >= end || *p < '0' || *p > '9') return fail("bad number");
            while (p < end && *p >= '0' && *p <= '9') p++;
        }
        e = p;
        return true;
    }
    bool skip() {                                                 // any value
        ws();
        if (p >= end) return fail("unexpected end of JSON input");
        const char c = *p;
        if (c == '"') return str(nullptr);
        if (c == '{' || c == '[') {
            if (++depth > 10000) return fail("exceeded max depth");
            struct Leave { int &d; ~Leave() { d--; } } leave{depth};
            if (c == '{') {
                p++; ws();
                if (p < end && *p == '}') { p++; return true; }
                for (;;) {
                    ws();
                    if (!str(nullptr)) return false;
                    ws();
                    if (p >= end || *p != ':') return fail("expected ':'");
                    p++;
                    if (!skip()) return false;
                    ws();
                    if (p < end && *p == ',') { p++; continue; }
                    if (p < end && *p == '}') { p++; return true; }
                    return fail("expected ',' or '}'");
                }
            }
            p++; ws();
            if (p < end && *p == ']') { p++; return true; }
            for (;;) {
                if (!skip()) return false;
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == ']') { p++; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (lit("true") || lit("false") || lit("null")) return true;
        const char *b, *e; bool integral;
        return num(b, e, integral);
    }
    // "key": value pairs of an object, one call of fn per pair (fn consumes the value)
    template <class F> bool object(F fn) {
        ws();
        if (p >= end || *p != '{') return fail("expected an object");
        p++; ws();
        if (p < end && *p == '}') { p++; return true; }
        for (;;) {
            ws();
            std::string key;
            if (!str(&key)) return false;
            ws();
            if (p >= end || *p != ':') return fail("expected ':'");
            p++; ws();
            if (!fn(key)) return false;
            ws();
            if (p < end && *p == ',') { p++; continue; }
            if (p < end && *p == '}') { p++; return true; }
            return fail("expected ',' or '}'");
        }
    }
    template <class F> bool array(F fn) {                         // fn consumes one element
        ws();
        if (p >= end || *p != '[') return fail("expected an array");
        p++; ws();
        if (p < end && *p == ']') { p++; return true; }
        for (;;) {
            ws();
            if (!fn()) return false;
            ws();
            if (p < end && *p == ',') { p++; continue; }
            if (p < end && *p == ']') { p++; return true; }
            return fail("expected ',' or ']'");
        }
    }
    // a string value into *out (sets *is_str); any other value is skipped
    bool str_value(std::string *out, bool *is_str) {
        ws();
        if (p < end && *p == '"') { out->clear(); *is_str = true; return str(out); }
        *is_str = false;
        return skip();
    }
    bool u64_value(uint64_t *out, bool *ok) {                     // a number that is a non-negative integer (as Go's uint fields take it)
        ws();
        *ok = false;
        if (p < end && (*p == '-' || (*p >= '0' && *p <= '9'))) {
            const char *b, *e; bool integral;
            if (!num(b, e, integral)) return false;
            if (integral && *b != '-' && e - b <= 20) {
                // (exact up to MaxUint64 — the value --khf signatures hold; the reference's own loader takes every number through a
                // float64 first, sketchio.go:115-116, and cannot read those back: this one is more lenient there, as the Python one was)
                unsigned __int128 v = 0;
                for (const char *q = b; q < e; q++) v = v * 10 + (unsigned)(*q - '0');
                if (v <= (unsigned __int128)0xFFFFFFFFFFFFFFFFull) { *out = (uint64_t)v; *ok = true; }
            } else {                                             // 1e3, 12.0 ...: the reference's first pass makes every number a float64
                const double dv = strtod(std::string(b, e).c_str(), nullptr);
                if (dv >= 0 && dv < 18446744073709551616.0 && dv == std::floor(dv)) { *out = (uint64_t)dv; *ok = true; }
            }
            return true;
        }
        return skip();
    }
};

bool ieq(const std::string &a, const char *b) {                   // encoding/json matches struct field names case-insensitively
    size_t i = 0;
    for (; i < a.size() && b[i]; i++) if (tolower((unsigned char)a[i]) != tolower((unsigned char)b[i])) return false;
    return i == a.size() && b[i] == 0;
}

bool parse_sketch_object(Reader &r, Sig &sg) {
    // (duplicate keys: the last one wins, as in a Go map)
    return r.object([&](const std::string &key) -> bool {
        if (ieq(key, "ksize")) { bool ok; uint64_t v = 0; if (!r.u64_value(&v, &ok)) return false; if (ok) sg.ksize = v; return true; }
        if (ieq(key, "md5sum")) { bool is; std::string v; if (!r.str_value(&v, &is)) return false; if (is) sg.md5 = v; return true; }
        if (ieq(key, "mins")) {
            sg.mins.clear();
            r.ws();
            if (r.p < r.end && *r.p != '[') return r.skip();      // null
            sg.mins.reserve(2048);
            return r.array([&]() -> bool {
                bool ok; uint64_t v = 0;
                if (!r.u64_value(&v, &ok)) return false;
                if (!ok) return r.fail("a value of \"mins\" is not an unsigned integer");
                sg.mins.push_back(v);
                return true;
            });
        }
        if (ieq(key, "weights")) {
            sg.weights.clear();
            r.ws();
            if (r.p < r.end && *r.p != '[') return r.skip();
            sg.weights.reserve(2048);
            return r.array([&]() -> bool {
                r.ws();
                const char *b, *e; bool integral;
                if (r.p >= r.end || !(*r.p == '-' || (*r.p >= '0' && *r.p <= '9'))) return r.fail("a value of \"weights\" is not a number");
                if (!r.num(b, e, integral)) return false;
                char tmp[64];
                const size_t n = (size_t)(e - b);
                double v;
                if (n < sizeof tmp) { memcpy(tmp, b, n); tmp[n] = 0; v = strtod(tmp, nullptr); }     // correctly rounded, as strconv.ParseFloat
                else v = strtod(std::string(b, e).c_str(), nullptr);
                sg.weights.push_back(v);
                return true;
            });
        }
        return r.skip();
    });
}

bool parse_doc(Reader &r, Doc &d) {
    const bool ok = r.object([&](const std::string &key) -> bool {
        static const char *names[6] = {"class", "filename", "hash_function", "license", "version", "banner_label"};
        std::string *dst[6] = {&d.cls, &d.filename, &d.hashfn, &d.license, &d.version, &d.banner};
        for (int i = 0; i < 6; i++)
            if (key == names[i]) { bool is; std::string v; if (!r.str_value(&v, &is)) return false; d.has[i] = is; if (is) *dst[i] = v; return true; }
        if (key == "signatures") {
            d.sigs.clear(); d.has_sigs = false;
            r.ws();
            if (r.p < r.end && *r.p != '[') return r.skip();
            d.has_sigs = true;
            return r.array([&]() -> bool {
                Sig sg;
                r.ws();
                if (r.p >= r.end || *r.p != '{') return r.fail("a signature is not an object");
                const bool oks = r.object([&](const std::string &k2) -> bool {
                    if (k2 == "Algorithm") { bool is; std::string v; if (!r.str_value(&v, &is)) return false; sg.has_algo = is; if (is) sg.algo = v; return true; }
                    if (k2 == "Sketch") {
                        r.ws();
                        if (r.p < r.end && *r.p == '{') { sg.has_sketch = true; sg.mins.clear(); sg.weights.clear(); sg.md5.clear(); sg.ksize = 0; return parse_sketch_object(r, sg); }
                        sg.has_sketch = false;
                        return r.skip();
                    }
                    return r.skip();
                });
                if (!oks) return false;
                d.sigs.push_back(std::move(sg));
                return true;
            });
        }
        return r.skip();
    });
    if (!ok) return false;
    r.ws();
    if (r.p != r.end) return r.fail("invalid character after top-level value");
    return true;
}

struct Loaded {                                                   // one file, after LoadHULKdata + FindSketch
    std::string path, error, banner;
    bool find_stage = false;                                      // the error is FindSketch's (raised in makeMatrix, after every file has loaded)
    std::vector<uint64_t> mins;
    std::vector<double> weights;
    bool histosketch = false;
};

bool read_file(const std::string &path, std::string &buf, std::string &err) {
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) { err = "open " + path + ": " + strerror(errno); return false; }
    struct stat st;
    if (fstat(fd, &st) != 0) { err = "stat " + path + ": " + strerror(errno); close(fd); return false; }
    buf.resize((size_t)st.st_size);
    size_t got = 0;
    while (got < buf.size()) {
        const ssize_t n = read(fd, &buf[got], buf.size() - got);
        if (n < 0) { if (errno == EINTR) continue; err = "read " + path + ": " + strerror(errno); close(fd); return false; }
        if (n == 0) break;
        got += (size_t)n;
    }
    close(fd);
    buf.resize(got);
    return true;
}

constexpr const char *HULK_VERSION = "1.0.0";                     // src/version/version.go

// LoadHULKdata (sketchio.go:100-195) + FindSketch (sketchio.go:198-257) for one file
void load_one(Loaded &L, uint64_t ksize, const std::string &algo) {
    std::string buf;
    if (!read_file(L.path, buf, L.error)) return;
    Doc d;
    Reader r{buf.data(), buf.data() + buf.size(), std::string()};
    if (!parse_doc(r, d)) { L.error = "malformed sketch file (" + r.err + "): " + L.path + "\n"; return; }
    static const char *names[6] = {"class", "filename", "hash_function", "license", "version", "banner_label"};
    for (int i = 0; i < 6; i++)                                   // (the reference's type assertions panic here)
        if (!d.has[i]) { L.error = std::string("malformed sketch file (\"") + names[i] + "\" is missing or not a string): " + L.path + "\n"; return; }
    if (!d.has_sigs) { L.error = "malformed sketch file (\"signatures\" is missing or not an array): " + L.path + "\n"; return; }
    for (const Sig &sg : d.sigs) {
        if (!sg.has_algo || !sg.has_sketch) { L.error = "malformed sketch file (a signature without \"Algorithm\" / \"Sketch\"): " + L.path + "\n"; return; }
        if (sg.algo != "histosketch" && sg.algo != "kmv" && sg.algo != "khf") { L.error = "unknown sketching algorithm: " + sg.algo; return; }
    }
    if (d.sigs.empty()) { L.error = "no signatures found in supplied file: " + L.path + "\n"; return; }
    if (d.cls != "hulk_sketch") { L.error = "JSON not created by HULK: " + L.path + "\n"; return; }
    if (d.version != HULK_VERSION) { L.error = "the loaded sketch was created with a different version of HULK: " + d.version + "\n"; return; }
    for (const Sig &sg : d.sigs) {
        if (sg.md5.empty()) { L.error = "no MD5 was stored for a sketch: " + d.filename + "\n"; return; }
        const std::string now = Md5::hex((const uint8_t *)sg.mins.data(), sg.mins.size() * 8);
        if (now != sg.md5) { L.error = "md5sum mismatch: " + sg.md5 + " vs. " + now + "\n"; return; }
    }
    // FindSketch
    L.find_stage = true;
    if (algo != "histosketch" && algo != "kmv" && algo != "khf") { L.error = "specified algorithm (" + algo + ") not found in the supplied sketch: " + d.filename + "\n"; return; }
    size_t with_algo = 0, hits = 0; const Sig *hit = nullptr;
    for (const Sig &sg : d.sigs) if (sg.algo == algo) { with_algo++; if (sg.ksize == ksize) { hits++; if (!hit) hit = &sg; } }
    if (!with_algo) { L.error = "no sketches were produced using the " + algo + " algorithm in file: " + d.filename + "\n"; return; }
    if (hits > 1) { L.error = "found " + std::to_string(hits) + " possible duplicate sketches in the supplied sketch file: " + d.filename + "\n"; return; }
    if (!hits) { L.error = "specified k-mer size (" + std::to_string(ksize) + ") not found in the supplied sketch file: " + d.filename + "\n"; return; }
    L.mins = hit->mins;
    L.histosketch = algo == "histosketch";
    if (L.histosketch) L.weights = hit->weights;
    L.banner = d.banner;
}

unsigned pick_threads(uint32_t threads, size_t jobs) {
    unsigned t = threads ? threads : std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
    if (t > 256) t = 256;
    if ((size_t)t > jobs) t = (unsigned)std::max<size_t>(1, jobs);
    return t;
}
template <class F> void parallel_for(size_t n, unsigned threads, F fn) {
    std::atomic<size_t> next{0};
    auto work = [&] { for (;;) { const size_t i = next.fetch_add(1); if (i >= n) return; fn(i); } };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < threads; t++) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
}

// encoding/csv Writer.fieldNeedsQuotes + the quoting of Writer.Write (UseCRLF false)
void csv_field(std::string &out, const std::string &f) {
    bool need = false;
    if (f == "\\.") need = true;
    else if (!f.empty()) {
        for (const char ch : f) if (ch == ',' || ch == '"' || ch == '\r' || ch == '\n') { need = true; break; }
        const unsigned char c0 = (unsigned char)f[0];
        if (c0 == ' ' || (c0 >= '\t' && c0 <= '\r')) need = true;                                      // unicode.IsSpace, ASCII
        if (c0 == 0xC2 && f.size() > 1 && ((unsigned char)f[1] == 0x85 || (unsigned char)f[1] == 0xA0)) need = true;   // U+0085, U+00A0
    }
    if (!need) { out += f; return; }
    out += '"';
    for (const char ch : f) { if (ch == '"') out += "\"\""; else out += ch; }
    out += '"';
}

void format_f2(std::string &out, double v) {                       // strconv.FormatFloat(v, 'f', 2, 64)
    if (v != v) { out += "NaN"; return; }
    if (std::isinf(v)) { out += v > 0 ? "+Inf" : "-Inf"; return; }
    char tmp[400];
    const int n = snprintf(tmp, sizeof tmp, "%.2f", v);
    out.append(tmp, (size_t)(n > 0 ? n : 0));
}

bool write_all(const std::string &path, const std::vector<std::string> &pieces, std::string &err) {
    const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) { err = "open " + path + ": " + strerror(errno); return false; }
    for (const std::string &s : pieces) {
        size_t at = 0;
        while (at < s.size()) {
            const ssize_t n = write(fd, s.data() + at, s.size() - at);
            if (n < 0) { if (errno == EINTR) continue; err = "write " + path + ": " + strerror(errno); close(fd); return false; }
            at += (size_t)n;
        }
    }
    close(fd);
    return true;
}

int put_err(char *errbuf, uint64_t errbuf_len, int code, const std::string &msg) {
    if (errbuf && errbuf_len) { const size_t n = std::min<size_t>(msg.size(), (size_t)errbuf_len - 1); memcpy(errbuf, msg.data(), n); errbuf[n] = 0; }
    fail(nullptr, code, msg);
    return code;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace
}  // namespace hulk

using namespace hulk;

struct hulk_sketch_set {
    std::vector<Loaded> files;                                    // sorted by path (sort.Strings: byte order)
    std::vector<uint64_t> mins;                                   // [n][size]
    std::vector<double> weights;                                  // [n][size] (zeros for the MinHash algorithms: they carry none)
    uint32_t size = 0;
    bool histosketch = false;
};

extern "C" {

int hulk_load_sketches(const char *const *paths, uint32_t n_paths, uint32_t ksize, const char *algo, uint32_t threads,
                       hulk_sketch_set **out, char *errbuf, uint64_t errbuf_len) {
    if (out) *out = nullptr;
    if (!out || !algo || (n_paths && !paths)) return put_err(errbuf, errbuf_len, HULK_ERR_ARG, "NULL");
    for (uint32_t i = 0; i < n_paths; i++) if (!paths[i]) return put_err(errbuf, errbuf_len, HULK_ERR_ARG, "NULL path");
    // cmd/smash.go:175-177 (hSketches is a map keyed by path: a path given twice counts once)
    std::vector<std::string> names(paths, paths + n_paths);
    std::sort(names.begin(), names.end());
    names.erase(std::unique(names.begin(), names.end()), names.end());
    hulk_sketch_set *set = new hulk_sketch_set;
    set->files.resize(names.size());
    for (size_t i = 0; i < names.size(); i++) set->files[i].path = names[i];
    const std::string algo_s = algo;
    parallel_for(names.size(), pick_threads(threads, names.size()), [&](size_t i) { load_one(set->files[i], ksize, algo_s); });
    // the reference loads the files in name order and dies at the first failure (cmd/smash.go:165-172), then wants two sketches
    // (:175-177); FindSketch failures surface in makeMatrix, pair by pair in sorted order: the first file that has one
    for (const Loaded &L : set->files)
        if (!L.error.empty() && !L.find_stage) { const std::string e = L.error; delete set; return put_err(errbuf, errbuf_len, HULK_ERR_ARG, e); }
    if (set->files.size() < 2) {
        const std::string e = std::to_string(set->files.size()) + " sketches found in the supplied directory, HULK needs at least 2 to smash!\n";
        delete set;
        return put_err(errbuf, errbuf_len, HULK_ERR_ARG, e);
    }
    for (const Loaded &L : set->files)
        if (!L.error.empty()) { const std::string e = L.error; delete set; return put_err(errbuf, errbuf_len, HULK_ERR_ARG, e); }
    const size_t size = set->files[0].mins.size();
    for (const Loaded &L : set->files)
        if (L.mins.size() != size) {
            const std::string e = "sketch length mismatch: " + std::to_string(size) + " vs " + std::to_string(L.mins.size()) + "\n";
            delete set;
            return put_err(errbuf, errbuf_len, HULK_ERR_ARG, e);
        }
    if (size > 0xffffffffull) { delete set; return put_err(errbuf, errbuf_len, HULK_ERR_ARG, "sketch too long"); }
    set->size = (uint32_t)size;
    set->histosketch = algo_s == "histosketch";
    const size_t n = set->files.size();
    set->mins.resize(n * size);
    set->weights.assign(n * size, 0.0);
    for (size_t i = 0; i < n; i++) {
        Loaded &L = set->files[i];
        if (size) memcpy(&set->mins[i * size], L.mins.data(), size * 8);
        if (set->histosketch) {
            // (a histosketch whose weights array is shorter than its mins: GetWJD indexes the weights by the mins' positions
            // and the reference panics; refused here)
            if (L.weights.size() != size) {
                const std::string e = "malformed sketch file (" + std::to_string(L.weights.size()) + " weights for " + std::to_string(size) + " mins): " + L.path + "\n";
                delete set;
                return put_err(errbuf, errbuf_len, HULK_ERR_ARG, e);
            }
            if (size) memcpy(&set->weights[i * size], L.weights.data(), size * 8);
        }
        std::vector<uint64_t>().swap(L.mins); std::vector<double>().swap(L.weights);
    }
    *out = set;
    return HULK_OK;
}

void hulk_sketch_set_free(hulk_sketch_set *set) { delete set; }

int hulk_sketch_set_info(const hulk_sketch_set *set, uint32_t *n_sketches, uint32_t *sketch_size) {
    if (!set) return HULK_ERR_ARG;
    if (n_sketches) *n_sketches = (uint32_t)set->files.size();
    if (sketch_size) *sketch_size = set->size;
    return HULK_OK;
}

const uint64_t *hulk_sketch_set_mins(const hulk_sketch_set *set) { return set ? set->mins.data() : nullptr; }
const double *hulk_sketch_set_weights(const hulk_sketch_set *set) { return set ? set->weights.data() : nullptr; }
const char *hulk_sketch_set_path(const hulk_sketch_set *set, uint32_t i) { return (set && i < set->files.size()) ? set->files[i].path.c_str() : nullptr; }
const char *hulk_sketch_set_banner(const hulk_sketch_set *set, uint32_t i) { return (set && i < set->files.size()) ? set->files[i].banner.c_str() : nullptr; }

int hulk_smash_files(int device, const char *const *paths, uint32_t n_paths, uint32_t ksize, const char *algo, const char *metric,
                     uint32_t threads, const char *matrix_csv_path, const char *banner_csv_path, double *distances,
                     hulk_smash_stats *stats, char *errbuf, uint64_t errbuf_len) {
    if (stats) memset(stats, 0, sizeof *stats);
    if (!algo || !metric) return put_err(errbuf, errbuf_len, HULK_ERR_ARG, "NULL");
    const std::string metric_s = metric, algo_s = algo;
    // cmd/smash.go:60-82
    if (metric_s != "jaccard" && metric_s != "weightedjaccard")
        return put_err(errbuf, errbuf_len, HULK_ERR_ARG, "supplied distance metric is not available: " + metric_s + "\nplease select one of the following: [jaccard weightedjaccard]");
    if (algo_s != "histosketch" && algo_s != "kmv" && algo_s != "khf")
        return put_err(errbuf, errbuf_len, HULK_ERR_ARG, "supplied algorithm not available: " + algo_s + "\nplease select one of the following: [histosketch kmv khf]");
    const double t0 = now_s();
    hulk_sketch_set *set = nullptr;
    { const int rc = hulk_load_sketches(paths, n_paths, ksize, algo, threads, &set, errbuf, errbuf_len); if (rc != HULK_OK) return rc; }
    struct Guard { hulk_sketch_set *s; ~Guard() { delete s; } } guard{set};
    const uint32_t n = (uint32_t)set->files.size(), S = set->size;
    if (metric_s == "weightedjaccard" && !set->histosketch)         // sketchio.go:287-293
        return put_err(errbuf, errbuf_len, HULK_ERR_ARG, "weighted jaccard is only supported for histosketches");
    const double t1 = now_s();
    std::vector<double> own;
    double *dist = distances;
    if (!dist) { own.resize((size_t)n * n); dist = own.data(); }
    double kernel_ms = 0.0;
    {
        const int rc = hulk_smash_ex(device, set->mins.data(), set->weights.data(), n, S, metric_s == "weightedjaccard" ? HULK_METRIC_WEIGHTED_JACCARD : HULK_METRIC_JACCARD,
                                     dist, &kernel_ms);
        if (rc != HULK_OK) return put_err(errbuf, errbuf_len, rc, hulk_last_error(nullptr));
    }
    const double t2 = now_s();
    if (matrix_csv_path) {
        std::vector<std::string> rows((size_t)n + 1);
        for (uint32_t i = 0; i < n; i++) { if (i) rows[0] += ','; csv_field(rows[0], set->files[i].path); }
        rows[0] += '\n';
        parallel_for(n, pick_threads(threads, n), [&](size_t s) {
            std::string &o = rows[s + 1];
            o.reserve((size_t)n * 7);
            for (uint32_t q = 0; q < n; q++) {
                if (q) o += ',';
                format_f2(o, 100 - (dist[s * (size_t)n + q] * 100));      // cmd/smash.go:217
            }
            o += '\n';
        });
        std::string err;
        if (!write_all(matrix_csv_path, rows, err)) return put_err(errbuf, errbuf_len, HULK_ERR_IO, err);
    }
    if (banner_csv_path) {                                          // makeBannerMatrix (cmd/smash.go:229-261), in sorted file order
        std::vector<std::string> rows(n);
        parallel_for(n, pick_threads(threads, n), [&](size_t i) {
            std::string &o = rows[i];
            for (uint32_t j = 0; j < S; j++) { o += std::to_string(set->mins[i * (size_t)S + j]); o += ','; }
            csv_field(o, set->files[i].banner);
            o += '\n';
        });
        std::string err;
        if (!write_all(banner_csv_path, rows, err)) return put_err(errbuf, errbuf_len, HULK_ERR_IO, err);
    }
    if (stats) {
        stats->seconds_load = t1 - t0; stats->seconds_matrix = t2 - t1; stats->seconds_csv = now_s() - t2; stats->kernel_ms = kernel_ms;
        stats->n_sketches = n; stats->sketch_size = S;
    }
    return HULK_OK;
}

}  // extern "C"
