// cws_gen.h — host generator of the CWS parameter stream of HistoSketch.newCWS
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

// go_rng GammaGenerator.Gamma(alpha>1, beta): Cheng (1977) rejection sampler as in CPython's
// random.gammavariate, with go_rng's squeeze constant 4*exp(-0.5)/sqrt(2).
class CwsGenerator {
  public:
    CwsGenerator() : g_(1), u_(1) {}          // DISTRIBUTION_SEED = 1 (histosketch.go:20)
    // one sketch slot: out[3*j + {0,1,2}] = {r, c, b} for bin j
    void next_row(double *out, size_t num_bins) {
        for (size_t j = 0; j < num_bins; j++) {
            const double r = gamma2();
            const double c = std::log(gamma2());
            const double b = (0.0 + u_.next_f64() * (1.0 - 0.0)) * r;
            out[3 * j] = r; out[3 * j + 1] = c; out[3 * j + 2] = b;
        }
    }

  private:
    double gamma2() {
        const double alpha = 2.0, beta = 1.0;
        static const double magic = 4 * std::exp(-0.5) / std::sqrt(2.0);
        static const double ainv = std::sqrt(2.0 * alpha - 1.0);
        static const double bbb = alpha - std::log(4.0);
        static const double ccc = alpha + ainv;
        for (;;) {
            const double u1 = g_.next_f64();
            if (!(1e-7 < u1 && u1 < .9999999)) continue;
            const double u2 = 1.0 - g_.next_f64();
            const double v = std::log(u1 / (1.0 - u1)) / ainv;
            const double x = alpha * std::exp(v);
            const double z = u1 * u1 * u2;
            const double rr = bbb + ccc * v - x;
            if (rr + magic - 4.5 * z >= 0.0 || rr >= std::log(z)) return x * beta;
        }
    }
    GoRandSource g_, u_;
};

}  // namespace hulk
