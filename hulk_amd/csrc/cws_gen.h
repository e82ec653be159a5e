// cws_gen.h — host half of the generator of the CWS parameter stream of HistoSketch.newCWS
// (reference src/histosketch/histosketch.go:95-126):
//     for slot { for bin { r = G.Gamma(2,1); c = ln(G.Gamma(2,1)); b = U.Float64Range(0,1) * r } }
// G = go_rng.NewGammaGenerator(1), U = go_rng.NewUniformGenerator(1); both wrap Go's math/rand
// additive lagged-Fibonacci source.  go_rng_cooked.h holds math/rand's seeding table, re-derived
// (not copied) by tools/derive_go_rngcooked.py.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>

#include "go_rng_cooked.h"

namespace hulk {

// math/rand rngSource: x[n] = x[n-607] + x[n-273] mod 2^64
class GoRandSource {
  public:
    explicit GoRandSource(int64_t seed) {
        tap_ = 0; feed_ = kLen - kTap;
        seed %= 2147483647;
        if (seed < 0) seed += 2147483647;
        if (seed == 0) seed = 89482311;
        int32_t x = (int32_t)seed;
        for (int i = -20; i < kLen; i++) {
            x = lcg(x);
            if (i >= 0) {
                uint64_t u = (uint64_t)x << 40;
                x = lcg(x); u ^= (uint64_t)x << 20;
                x = lcg(x); u ^= (uint64_t)x;
                vec_[i] = u ^ GO_RNG_COOKED[i];
            }
        }
    }
    // the 607 sequence values y[-606..0] a freshly seeded source starts from (y[1] is the first next_u64()); only
    // valid before the first draw
    void initial_window(uint64_t out[607]) const {
        for (int j = 0; j < kLen; j++) out[j] = vec_[((kLen - kTap) + (kLen - 1) - j + kLen) % kLen];
    }
    inline uint64_t next_u64() {
        if (--tap_ < 0) tap_ += kLen;
        if (--feed_ < 0) feed_ += kLen;
        return vec_[feed_] += vec_[tap_];
    }
    // Rand.Float64: float64(Int63()) / (1<<63), resampled when it rounds to 1
    inline double next_f64() {
        for (;;) {
            const double f = (double)(int64_t)(next_u64() & 0x7fffffffffffffffull) * 0x1p-63;
            if (f != 1.0) return f;
        }
    }

  private:
    static constexpr int kLen = 607, kTap = 273;
    static int32_t lcg(int32_t x) {            // 48271 * x mod (2^31 - 1), Schrage
        const int32_t hi = x / 44488, lo = x % 44488;
        x = 48271 * lo - 3399 * hi;
        return x < 0 ? x + 2147483647 : x;
    }
    uint64_t vec_[kLen];
    int tap_, feed_;
};

// go_rng GammaGenerator.Gamma(alpha > 1, beta): Cheng (1977) rejection sampler as in CPython's
// random.gammavariate, with go_rng's squeeze constant 4*exp(-0.5)/sqrt(2):
//     loop { u1 = U();  if !(1e-7 < u1 < .9999999) continue;  u2 = 1 - U();  ...accept/reject... }
// The uniform stream is sequential, the transcendental math is not: the host only walks the raw
// stream and pairs every attempt's (u1, u2) — integer compares, ~1 ns per value — and the GPU
// evaluates all attempts of a chunk in parallel and compacts the accepted values in order
// (k_cws_eval / k_cws_scatter in hulk_cws.hip).
struct CwsConstants {
    double ainv, bbb, ccc, magic;            // alpha = 2, beta = 1 (histosketch.go:112-113)
    // cpython_squeeze: 1 + ln 4.5 (CPython's SG_MAGICCONST) instead of go_rng's recalled 4*exp(-0.5)/sqrt(2) — see
    // HULK_FLAG_GAMMA_CPYTHON in include/hulk_hip.h
    explicit CwsConstants(bool cpython_squeeze = false) {
        const double alpha = 2.0;
        magic = cpython_squeeze ? 1.0 + std::log(4.5) : 4 * std::exp(-0.5) / std::sqrt(2.0);
        ainv = std::sqrt(2.0 * alpha - 1.0);
        bbb = alpha - std::log(4.0);
        ccc = alpha + ainv;
    }
};

class AttemptStream {
  public:
    AttemptStream() : g_(1) {}                // gamma generator's own uniform source, seed 1
    // fills `pairs` with the raw int63 values (u1, u2) of the next `n` attempts that pass the
    // u1 range test; Float64()'s "== 1, resample" rule is applied here too
    void fill(uint64_t *pairs, size_t n) {
        for (size_t i = 0; i < n; i++) {
            uint64_t a;
            for (;;) { a = next63(); const double u1 = (double)(int64_t)a * 0x1p-63; if (1e-7 < u1 && u1 < .9999999) break; }
            pairs[2 * i] = a;
            pairs[2 * i + 1] = next63();
        }
    }

  private:
    uint64_t next63() {                       // Rand.Float64's input, skipping values that round to 1.0
        for (;;) {
            const uint64_t x = g_.next_u64() & 0x7fffffffffffffffull;
            if ((double)(int64_t)x * 0x1p-63 != 1.0) return x;
        }
    }
    GoRandSource g_;
};

// the separate uniform generator (seed 1) of newCWS: one Float64 per table entry
class UniformStream {
  public:
    UniformStream() : u_(1) {}
    void fill(uint64_t *raw, size_t n) {
        for (size_t i = 0; i < n; i++) {
            for (;;) {
                const uint64_t x = u_.next_u64() & 0x7fffffffffffffffull;
                if ((double)(int64_t)x * 0x1p-63 != 1.0) { raw[i] = x; break; }
            }
        }
    }

  private:
    GoRandSource u_;
};

}  // namespace hulk
