// crc32_clmul.h — the gzip CRC-32 by carry-less multiplication (x86-64 PCLMULQDQ), for the ingest path.
//
// zlib 1.2.11's crc32 (slicing by 4) runs at 1-2 GB/s on one core — as much time as inflating the text it checks.  Folding
// 64 bytes per step with four independent 128-bit accumulators (Gopal et al., "Fast CRC Computation for Generic Polynomials
// Using PCLMULQDQ Instruction", Intel 2009; the constants are x^(n) mod P for the reflected polynomial 0xEDB88320) does
// 10+ GB/s.  `hulk::crc32_fast` has zlib's signature and semantics; it uses the folding code when the CPU has PCLMULQDQ +
// SSE4.1 AND a self-test against zlib's crc32 (run once, lengths 0..700 at odd alignments, chained calls) agreed — otherwise
// it IS zlib's crc32.
#pragma once
#include <stdint.h>
#include <string.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace hulk {

#if defined(__x86_64__)
// `crc` in zlib's convention (as returned by crc32()); n >= 64 and a multiple of 16
__attribute__((target("pclmul,sse4.1"))) static inline uint32_t crc32_fold(uint32_t crc, const uint8_t *p, size_t n) {
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll);      // fold by 512 bits
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);      // fold by 128 bits
    const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124ll);                      // 96 -> 64 bits
    const __m128i poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);       // Barrett: P', mu
    __m128i x1 = _mm_loadu_si128((const __m128i *)(p + 0)), x2 = _mm_loadu_si128((const __m128i *)(p + 16));
    __m128i x3 = _mm_loadu_si128((const __m128i *)(p + 32)), x4 = _mm_loadu_si128((const __m128i *)(p + 48));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)~crc));
    p += 64; n -= 64;
    while (n >= 64) {
        const __m128i a1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), a2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
        const __m128i a3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), a4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11); x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11); x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, a1), _mm_loadu_si128((const __m128i *)(p + 0)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, a2), _mm_loadu_si128((const __m128i *)(p + 16)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, a3), _mm_loadu_si128((const __m128i *)(p + 32)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, a4), _mm_loadu_si128((const __m128i *)(p + 48)));
        p += 64; n -= 64;
    }
    // four accumulators into one
    __m128i a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), a), x2);
    a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), a), x3);
    a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), a), x4);
    while (n >= 16) {
        a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
        x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), a), _mm_loadu_si128((const __m128i *)p));
        p += 16; n -= 16;
    }
    // 128 -> 64 bits
    const __m128i mask32 = _mm_setr_epi32(~0, 0, ~0, 0);
    __m128i t = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
    t = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, mask32);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(x1, k5, 0x00), t);
    // Barrett reduction to 32 bits
    t = _mm_and_si128(x1, mask32);
    t = _mm_clmulepi64_si128(t, poly, 0x10);
    t = _mm_and_si128(t, mask32);
    t = _mm_clmulepi64_si128(t, poly, 0x00);
    x1 = _mm_xor_si128(x1, t);
    return ~(uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

static inline uint32_t crc32_zlib(uint32_t crc, const uint8_t *p, size_t n) {
    for (size_t at = 0; at < n; at += 1u << 30) crc = (uint32_t)::crc32(crc, p + at, (uInt)(n - at < (1u << 30) ? n - at : (1u << 30)));
    return crc;
}

#if defined(__x86_64__)
static inline uint32_t crc32_fast_unchecked(uint32_t crc, const uint8_t *p, size_t n) {
    if (n >= 64) { const size_t m = n & ~(size_t)15; crc = crc32_fold(crc, p, m); p += m; n -= m; }
    return n ? crc32_zlib(crc, p, n) : crc;
}
static inline bool crc32_fast_usable() {
    static const bool ok = [] {
        if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return false;
        uint8_t buf[800];
        uint32_t s = 12345;
        for (auto &b : buf) { s = s * 1664525u + 1013904223u; b = (uint8_t)(s >> 24); }
        for (size_t off = 0; off < 3; off++)
            for (size_t n = 0; n + off <= 700; n++)
                if (crc32_fast_unchecked(0, buf + off, n) != crc32_zlib(0, buf + off, n)) return false;
        // chained: the running value goes in as zlib's does
        uint32_t a = 0, b = 0;
        for (size_t at = 0; at < 700; at += 100) { a = crc32_fast_unchecked(a, buf + at, 100); b = crc32_zlib(b, buf + at, 100); }
        return a == b;
    }();
    return ok;
}
#endif

static inline uint32_t crc32_fast(uint32_t crc, const uint8_t *p, size_t n) {
#if defined(__x86_64__)
    if (crc32_fast_usable()) return crc32_fast_unchecked(crc, p, n);
#endif
    return crc32_zlib(crc, p, n);
}

}  // namespace hulk
