/*
 * hulk_oracle.c — CPU restatement of the will-rowe/hulk v1.0.0 `sketch` hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke()
 * and bench.py's `cpu_baseline` leg may load it; hulk_amd/ never does.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference repository root).  The shape is deliberately literal (per-read set,
 * per-minimizer jump hash, per-bin AddElement with 2 exp + 1 log over three
 * [slot][bin] fp64 matrices) so that it doubles as the CPU baseline.
 *
 * PARITY STATUS ("what pins this oracle"):
 *   - nt4 table, Pow, kmerspectrum cardinalities, deque behaviour: pinned by the
 *     reference's own unit tests (src/minimizer/minimizer_test.go:14-30,
 *     src/kmerspectrum/kmerspectrum_test.go:14-45, src/queue/queue_test.go:7-30,
 *     src/helpers/helpers_test.go:7-17) — see tests/test_oracle_reference_vectors.py.
 *   - jump hash: third-party github.com/dgryski/go-jump @ e1f439676b57 (not vendored in
 *     the reference).  Restated from the published Lamping–Veach algorithm
 *     (arXiv:1406.2294); pinned by that package's published test vectors.
 *   - Go math/rand (stdlib, seed 1): pinned — the rngCooked table is derived by
 *     tools/derive_go_rngcooked.py and reproduces the known Seed(1) output stream.
 *   - github.com/leesper/go_rng @ a612b043e353 Gamma/Uniform: restated from the
 *     published algorithm (a port of CPython random.gammavariate, Cheng 1977) —
 *     source NOT available offline, and the reference has NO test that pins any CWS
 *     value, sketch `mins` or `weights`  =>  **CWS parity with Go-produced sketches is
 *     UNPINNED** (everything upstream of it — minimizers, bins, histogram, CMS — is
 *     integer-exact and pinned as above).  What IS pinned: the restatement equals CPython's
 *     own random.gammavariate bit for bit when both are fed Go's math/rand Float64 stream
 *     (tests/test_oracle_reference_vectors.py::test_gamma_matches_cpython_gammavariate_on_the_go_stream),
 *     and the squeeze constant — go_rng's one doubtful detail — cannot change the stream
 *     (::test_gamma_squeeze_constant_is_immaterial).  What remains an assumption is that
 *     go_rng IS that port, call for call (u1 range test, u2 = 1 - U(), one shared source).
 *   - The reference cannot be built here (no Go toolchain, deps not vendored), so there
 *     is no oracle/_ref.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#include "go_rng_cooked.h"

#define ORC_OK 0
#define ORC_ERR_W -1        /* "w must be: 0 < w < 257"                     minimizer.go:62-64 */
#define ORC_ERR_K -2        /* "k size must be: 0 < k < 32"                 minimizer.go:65-67 */
#define ORC_ERR_EMPTY -3    /* "sequence length must be > 0"                minimizer.go:71-73 */
#define ORC_ERR_SHORT -4    /* "sequence length must be >= w + k - 1"       minimizer.go:74-76 */
#define ORC_ERR_FEWBINS -5  /* "not used yet" (<1 % of bins used)           kmerspectrum.go:94-96 */
#define ORC_ERR_HS_K -6     /* "histosketching only supports k <= 31"       histosketch.go:53-55 */
#define ORC_ERR_DECAY -7    /* "decay ratio must be between 0.0 and 1.0"    histosketch.go:62-64 */
#define ORC_ERR_BINS -8     /* "histogram must have at least 2 bins"        histosketch.go:65-67 */
#define ORC_ERR_NEGBINS -9  /* "negative value used for number of k-mer spectrum bins" kmerspectrum.go:33-35 */
#define ORC_ERR_NOSEQ -10   /* "no sequences received"                      pipeline/sketch.go:237-239 */
#define ORC_ERR_ALLOC -20

/* ------------------------------------------------------------------ nt4 table
 * src/minimizer/minimizer.go:13-30: A/a=0 C/c=1 G/g=2 T/t/U/u=3, bytes 0..3 map to
 * themselves, everything else 4. */
static uint8_t NT4[256];
static int nt4_ready = 0;
static void nt4_init(void) {
    if (nt4_ready) return;
    memset(NT4, 4, sizeof NT4);
    NT4[0] = 0; NT4[1] = 1; NT4[2] = 2; NT4[3] = 3;
    NT4['A'] = NT4['a'] = 0;
    NT4['C'] = NT4['c'] = 1;
    NT4['G'] = NT4['g'] = 2;
    NT4['T'] = NT4['t'] = 3;
    NT4['U'] = NT4['u'] = 3;
    nt4_ready = 1;
}
uint8_t orc_nt4(uint8_t c) { nt4_init(); return NT4[c]; }

/* ------------------------------------------------------------------ hash64
 * src/minimizer/minimizer.go:33-42 (minimap2 invertible integer hash). */
uint64_t orc_hash64(uint64_t key, uint64_t mask) {
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

/* ------------------------------------------------------------------ Pow
 * src/helpers/helpers.go:18-28 (integer power; bins = Pow(k,4), cmd/sketch.go:118). */
uint64_t orc_pow(uint64_t a, uint64_t b) {
    uint64_t p = 1;
    while (b > 0) {
        if (b & 1) p *= a;
        b >>= 1;
        a *= a;
    }
    return p;
}

/* ------------------------------------------------------------------ jump hash
 * github.com/dgryski/go-jump Hash(key uint64, numBuckets int) int32, called at
 * src/kmerspectrum/kmerspectrum.go:70 and src/countmin/countmin.go:125.
 * Lamping & Veach, "A Fast, Minimal Memory, Consistent Hash Algorithm". */
int32_t orc_jump(uint64_t key, int64_t num_buckets) {
    int64_t b = -1, j = 0;
    if (num_buckets <= 0) num_buckets = 1;
    while (j < num_buckets) {
        b = j;
        key = key * 2862933555777941757ULL + 1;
        j = (int64_t)((double)(b + 1) * ((double)(1LL << 31) / (double)((key >> 33) + 1)));
    }
    return (int32_t)b;
}

/* ------------------------------------------------------------------ minimizers
 * src/minimizer/minimizer.go:59-93 (NewMinimizerSketch) + :96-204 (findMinimizers),
 * deque semantics from src/queue/queue.go:6-63, per-read set semantics from
 * golang-set (Add/Contains; order irrelevant).  Writes the distinct minimizer values
 * of the read to out[] in first-emission order; returns their count or an ORC_ERR_*. */
typedef struct { uint64_t X; int32_t Y; } orc_pair;    /* queue.Pair, queue.go:6-9 */

int32_t orc_minimizers(const uint8_t *seq, int32_t seq_len, uint32_t k_, uint32_t w_,
                       uint64_t *out, int32_t cap) {
    nt4_init();
    if (w_ > 256) return ORC_ERR_W;
    if (k_ > 31) return ORC_ERR_K;
    if (seq_len < 1) return ORC_ERR_EMPTY;
    if (seq_len < (int32_t)(w_ + k_ - 1)) return ORC_ERR_SHORT;
    const int32_t k = (int32_t)k_, w = (int32_t)w_;

    uint64_t kmers[2] = {0, 0};
    int32_t kmerSpan = 0;
    const uint64_t bitmask = ((uint64_t)1 << (uint64_t)(2 * k)) - 1;
    const uint64_t bitshift = (uint64_t)(2 * (k - 1));

    /* the deque: at most seq_len live pairs */
    orc_pair *q = (orc_pair *)malloc(sizeof(orc_pair) * (size_t)(seq_len + 1));
    if (!q) return ORC_ERR_ALLOC;
    int32_t qh = 0, qt = 0;                 /* [qh, qt) */
    int32_t nset = 0;

    for (int32_t i = 0; i < seq_len; i++) {
        const int32_t windowIndex = i - w + 1;
        const uint8_t c = NT4[seq[i]];
        /* :118-122 — `if c > 3 {}` is empty in the reference: N is NOT special-cased */
        if ((windowIndex + 1) < k) kmerSpan = windowIndex + 1; else kmerSpan = k;
        kmers[0] = (kmers[0] << 2 | (uint64_t)c) & bitmask;
        kmers[1] = (kmers[1] >> 2) | ((uint64_t)3 ^ (uint64_t)c) << bitshift;   /* never masked */
        if (i < k - 1) continue;
        if (kmers[0] == kmers[1]) continue;
        unsigned strand = 0;
        if (kmers[0] > kmers[1]) strand = 1;
        orc_pair cur;
        /* Go: uint64(int32) sign-extends */
        cur.X = orc_hash64(kmers[strand], bitmask) << 8 | (uint64_t)(int64_t)kmerSpan;
        cur.Y = i;
        if (qt != qh) {
            for (;;) {
                if (qt == qh || q[qh].Y > (i - w)) break;
                qh++;
            }
            for (;;) {
                if (qt == qh || q[qt - 1].X < cur.X) break;
                qt--;
            }
        }
        if (qt == qh) { qh = qt = 0; }      /* reuse storage; semantics unchanged */
        q[qt++] = cur;
        if (windowIndex >= 0) {
            const uint64_t m = q[qh].X;
            int found = 0;
            for (int32_t s = 0; s < nset; s++) if (out[s] == m) { found = 1; break; }
            if (!found) {
                if (nset >= cap) { free(q); return ORC_ERR_ALLOC; }
                out[nset++] = m;
            }
        }
    }
    free(q);
    return nset;
}

/* ------------------------------------------------------------------ Go math/rand
 * stdlib math/rand rngSource (Seed / Uint64 / Int63) and Rand.Float64, as used by
 * go_rng's generators (rand.New(rand.NewSource(seed))).  Table: go_rng_cooked.h. */
#define GO_RNG_LEN 607
#define GO_RNG_TAP 273
typedef struct { uint64_t vec[GO_RNG_LEN]; int tap, feed; } orc_gosrc;

static int32_t go_seedrand(int32_t x) {
    const int32_t A = 48271, Q = 44488, R = 3399;
    int32_t hi = x / Q, lo = x % Q;
    x = A * lo - R * hi;
    if (x < 0) x += 2147483647;
    return x;
}
void orc_gosrc_seed(orc_gosrc *s, int64_t seed) {
    s->tap = 0; s->feed = GO_RNG_LEN - GO_RNG_TAP;
    seed = seed % 2147483647;
    if (seed < 0) seed += 2147483647;
    if (seed == 0) seed = 89482311;
    int32_t x = (int32_t)seed;
    for (int i = -20; i < GO_RNG_LEN; i++) {
        x = go_seedrand(x);
        if (i >= 0) {
            int64_t u = (int64_t)x << 40;
            x = go_seedrand(x); u ^= (int64_t)x << 20;
            x = go_seedrand(x); u ^= (int64_t)x;
            s->vec[i] = (uint64_t)u ^ GO_RNG_COOKED[i];
        }
    }
}
uint64_t orc_gosrc_uint64(orc_gosrc *s) {
    if (--s->tap < 0) s->tap += GO_RNG_LEN;
    if (--s->feed < 0) s->feed += GO_RNG_LEN;
    uint64_t x = s->vec[s->feed] + s->vec[s->tap];
    s->vec[s->feed] = x;
    return x;
}
int64_t orc_gosrc_int63(orc_gosrc *s) { return (int64_t)(orc_gosrc_uint64(s) & 0x7fffffffffffffffULL); }
double orc_gosrc_float64(orc_gosrc *s) {
    for (;;) {
        double f = (double)orc_gosrc_int63(s) / 9223372036854775808.0;
        if (f == 1.0) continue;              /* Rand.Float64 resamples */
        return f;
    }
}
orc_gosrc *orc_gosrc_new(int64_t seed) {
    orc_gosrc *s = (orc_gosrc *)malloc(sizeof *s);
    if (s) orc_gosrc_seed(s, seed);
    return s;
}
void orc_gosrc_free(orc_gosrc *s) { free(s); }

/* ------------------------------------------------------------------ go_rng
 * github.com/leesper/go_rng: UniformGenerator.Float64Range(a,b) = a + Float64()*(b-a);
 * GammaGenerator.Gamma(alpha,beta) for alpha > 1 — Cheng's GB rejection sampler as in
 * CPython random.gammavariate, with go_rng's MAGIC_CONST = 4*exp(-0.5)/sqrt(2).
 * Call sites: src/histosketch/histosketch.go:103-104,112,113,116.  [restated; unpinned] */
double orc_go_uniform_range(orc_gosrc *s, double a, double b) {
    return a + orc_gosrc_float64(s) * (b - a);
}
/* The squeeze constant is the one uncertain recollection of go_rng: go_rng is believed to define
 * MAGIC_CONST = 4*exp(-0.5)/sqrt(2) (variant 0, default); CPython's gammavariate uses SG_MAGICCONST = 1 + ln 4.5
 * (variant 1); variant 2 has no squeeze at all.  It turns out NOT to matter: `r + M - 4.5z >= 0` is only a shortcut
 * that implies `r >= ln z` for every M <= 1 + ln 4.5 (tangent bound ln z <= 4.5z - 1 - ln 4.5), so the accept/reject
 * decision — and with it the whole parameter stream — is the same for all three (tests/test_oracle_reference_vectors.py
 * ::test_gamma_squeeze_constant_is_immaterial checks 10^6 draws).  Product counterpart: HULK_FLAG_GAMMA_CPYTHON. */
static int g_gamma_variant = 0;
void orc_set_gamma_variant(int v) { g_gamma_variant = v; }
int orc_get_gamma_variant(void) { return g_gamma_variant; }
double orc_go_gamma(orc_gosrc *s, double alpha, double beta) {
    const double MAGIC_CONST = g_gamma_variant == 2 ? -HUGE_VAL : g_gamma_variant == 1 ? 1.0 + log(4.5) : 4 * exp(-0.5) / sqrt(2.0);
    /* only the alpha > 1 branch is reachable from HULK (Gamma(2,1)) */
    const double ainv = sqrt(2.0 * alpha - 1.0);
    const double bbb = alpha - log(4.0);
    const double ccc = alpha + ainv;
    for (;;) {
        double u1 = orc_gosrc_float64(s);
        if (!(1e-7 < u1 && u1 < .9999999)) continue;
        double u2 = 1.0 - orc_gosrc_float64(s);
        double v = log(u1 / (1.0 - u1)) / ainv;
        double x = alpha * exp(v);
        double z = u1 * u1 * u2;
        double r = bbb + ccc * v - x;
        if (r + MAGIC_CONST - 4.5 * z >= 0.0 || r >= log(z)) return x * beta;
    }
}

/* newCWS — src/histosketch/histosketch.go:95-126.  Slot-major, bin-minor:
 * r = Gamma(2,1); c = ln(Gamma(2,1)) (same gamma generator, seed 1);
 * b = U(0,1)*r (separate uniform generator, seed 1).  Arrays are [slot][bin] row-major. */
#define DISTRIBUTION_SEED 1   /* histosketch.go:20 */
int orc_cws_fill(uint32_t S, int32_t B, double *r, double *c, double *b) {
    orc_gosrc g, u;
    orc_gosrc_seed(&g, DISTRIBUTION_SEED);
    orc_gosrc_seed(&u, DISTRIBUTION_SEED);
    for (uint32_t i = 0; i < S; i++)
        for (int32_t j = 0; j < B; j++) {
            size_t o = (size_t)i * (size_t)B + (size_t)j;
            r[o] = orc_go_gamma(&g, 2, 1);
            c[o] = log(orc_go_gamma(&g, 2, 1));
            b[o] = orc_go_uniform_range(&u, 0, 1) * r[o];
        }
    return ORC_OK;
}

/* ------------------------------------------------------------------ count-min
 * src/countmin/countmin.go:11-57 (constructor), :103-147 (Add/traverse/scale). */
#define CMS_EPSILON 0.001
#define CMS_DELTA 0.99
typedef struct {
    uint32_t depth, width;
    double *ctr;                 /* [depth][width] */
    int applyScaling;
    double decayWeight;
} orc_cms;

static int cms_init(orc_cms *q, double decayRatio) {
    q->width = (uint32_t)ceil(2 / CMS_EPSILON);
    q->depth = (uint32_t)ceil(log(1 - CMS_DELTA) / log(0.5));
    q->ctr = (double *)calloc((size_t)q->depth * q->width, sizeof(double));
    if (!q->ctr) return ORC_ERR_ALLOC;
    q->decayWeight = 0.0;
    if (decayRatio > 0.0 && decayRatio < 1.0) {
        q->decayWeight = exp(-decayRatio);
        q->applyScaling = 1;
    } else {
        q->applyScaling = 0;
    }
    return ORC_OK;
}
static double cms_add(orc_cms *q, uint64_t element, double increment) {
    if (q->applyScaling) {
        for (uint32_t d = 0; d < q->depth; d++)
            for (uint32_t g = 0; g < q->width; g++)
                q->ctr[d * q->width + g] = q->ctr[d * q->width + g] * q->decayWeight;
    }
    double currentMinimum = DBL_MAX;
    for (uint32_t d = 0; d < q->depth; d++) {
        uint64_t hash = element + ((uint64_t)d * element);
        int32_t g = orc_jump(hash, (int64_t)q->width);
        double *p = &q->ctr[d * q->width + (uint32_t)g];
        if (increment != 0.0) *p += increment;
        if (*p < currentMinimum) currentMinimum = *p;
    }
    return currentMinimum;
}
/* exposed so tests can pin the geometry and the per-row positions */
void orc_cms_geometry(uint32_t *depth, uint32_t *width) {
    orc_cms q; if (cms_init(&q, 1.0) == ORC_OK) { *depth = q.depth; *width = q.width; free(q.ctr); }
}

/* ------------------------------------------------------------------ the sketcher
 * kmerspectrum (src/kmerspectrum/kmerspectrum.go:30-112), histosketch
 * (src/histosketch/histosketch.go:50-155), and the interval / flush logic of
 * SeqMinimizer.Run + boss (src/pipeline/sketch.go:196-224, src/pipeline/boss.go:90-128)
 * with the deterministic reading "interval t = reads [tI,(t+1)I)" (the reference races
 * at flush boundaries — boss.go:114 TODO). */
typedef struct orc_sketcher {
    uint32_t k, w, S;
    int32_t B;
    uint32_t interval;
    double decayRatio;
    /* kmerspectrum */
    double *bins;                /* float64 counters, kmerspectrum.go:25 */
    int32_t used;                /* bitvector popcount stand-in */
    /* histosketch */
    uint64_t *sketch;            /* `mins`   */
    double *weights;             /* `weights` */
    int applyConceptDrift;
    double *r, *c, *b;           /* CWS [slot][bin] */
    orc_cms cms;
    /* counters (pipeline/sketch.go:186-208, boss.go:93) */
    uint64_t seqCount, lengthTotal, minimizerCounter, flushes, elements;
    uint64_t *scratch; int32_t scratch_cap;
} orc_sketcher;

void orc_free(orc_sketcher *o) {
    if (!o) return;
    free(o->bins); free(o->sketch); free(o->weights);
    free(o->r); free(o->c); free(o->b); free(o->cms.ctr); free(o->scratch);
    free(o);
}

/* num_bins <= 0 selects the CLI's Pow(k,4) (cmd/sketch.go:118). */
int orc_new(uint32_t k, uint32_t w, uint32_t S, int32_t num_bins, double decayRatio,
            uint32_t interval, orc_sketcher **out) {
    *out = NULL;
    if (num_bins == 0) num_bins = (int32_t)orc_pow(k, 4);
    if (num_bins < 0) return ORC_ERR_NEGBINS;
    if (k > 31) return ORC_ERR_HS_K;
    if (decayRatio < 0.0 || decayRatio > 1.0) return ORC_ERR_DECAY;
    if (num_bins < 2) return ORC_ERR_BINS;
    orc_sketcher *o = (orc_sketcher *)calloc(1, sizeof *o);
    if (!o) return ORC_ERR_ALLOC;
    o->k = k; o->w = w; o->S = S; o->B = num_bins; o->interval = interval; o->decayRatio = decayRatio;
    size_t n = (size_t)S * (size_t)num_bins;
    o->bins = (double *)calloc((size_t)num_bins, sizeof(double));
    o->sketch = (uint64_t *)calloc(S ? S : 1, sizeof(uint64_t));
    o->weights = (double *)malloc((S ? S : 1) * sizeof(double));
    o->r = (double *)malloc((n ? n : 1) * sizeof(double));
    o->c = (double *)malloc((n ? n : 1) * sizeof(double));
    o->b = (double *)malloc((n ? n : 1) * sizeof(double));
    o->scratch_cap = 1 << 16;
    o->scratch = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)o->scratch_cap);
    if (!o->bins || !o->sketch || !o->weights || !o->r || !o->c || !o->b || !o->scratch ||
        cms_init(&o->cms, decayRatio) != ORC_OK) { orc_free(o); return ORC_ERR_ALLOC; }
    if (decayRatio != 1.0) o->applyConceptDrift = 1;
    for (uint32_t i = 0; i < S; i++) { o->sketch[i] = 0; o->weights[i] = DBL_MAX; }
    orc_cws_fill(S, num_bins, o->r, o->c, o->b);
    *out = o;
    return ORC_OK;
}

/* CWS.getSample — histosketch.go:30-33 */
static inline double cws_sample(const orc_sketcher *o, uint64_t i, uint32_t j, double freq) {
    size_t at = (size_t)j * (size_t)o->B + (size_t)i;
    double Yka = exp(log(freq) - o->b[at]);
    return o->c[at] / (Yka * exp(o->r[at]));
}

/* HistoSketch.AddElement — histosketch.go:129-155 */
void orc_add_element(orc_sketcher *o, uint64_t bin, double value) {
    double estiFreq = cms_add(&o->cms, bin, value);
    for (uint32_t slot = 0; slot < o->S; slot++) {
        double Aka = cws_sample(o, bin, slot, estiFreq);
        double curMin;
        if (o->applyConceptDrift) curMin = o->weights[slot] / o->cms.decayWeight;
        else curMin = o->weights[slot];
        if (Aka < curMin) { o->sketch[slot] = bin; o->weights[slot] = Aka; }
    }
    o->elements++;
}

/* KmerSpectrum.AddHash — kmerspectrum.go:67-81 */
static void ks_add_hash(orc_sketcher *o, uint64_t kmer) {
    int32_t bin = orc_jump(kmer, (int64_t)o->B);
    if (o->bins[bin] == 0.0) o->used++;
    o->bins[bin]++;
}

/* boss flush (boss.go:112-128) = Cardinality/Dump/Wipe (kmerspectrum.go:53-64,84-112) */
int orc_flush(orc_sketcher *o) {
    if (o->used == 0) return ORC_OK;
    double propUsed = (double)o->used / (double)o->B;
    if (propUsed < 0.01) return ORC_ERR_FEWBINS;
    for (int32_t i = 0; i < o->B; i++)
        if (o->bins[i] != 0.0) orc_add_element(o, (uint64_t)i, o->bins[i]);
    for (int32_t i = 0; i < o->B; i++) o->bins[i] = 0;
    o->used = 0;
    o->flushes++;
    return ORC_OK;
}

/* KmerSpectrum.Wipe alone (kmerspectrum.go:53-64): used by bench.py's all-cores CPU leg, whose worker threads only
 * bin reads and hand their spectrum to the one sketching thread */
void orc_wipe(orc_sketcher *o) {
    for (int32_t i = 0; i < o->B; i++) o->bins[i] = 0;
    o->used = 0;
}

/* one iteration of SeqMinimizer.Run's loop (pipeline/sketch.go:197-215) +
 * Minion (minion.go:45-57) + collector (boss.go:90-95) */
int orc_add_read(orc_sketcher *o, const uint8_t *seq, int32_t len) {
    if (len > o->scratch_cap) {
        uint64_t *n = (uint64_t *)realloc(o->scratch, sizeof(uint64_t) * (size_t)len);
        if (!n) return ORC_ERR_ALLOC;
        o->scratch = n; o->scratch_cap = len;
    }
    int32_t n = orc_minimizers(seq, len, o->k, o->w, o->scratch, o->scratch_cap);
    if (n < 0) return n;
    for (int32_t i = 0; i < n; i++) { ks_add_hash(o, o->scratch[i]); o->minimizerCounter++; }
    o->seqCount++;
    o->lengthTotal += (uint64_t)len;
    if (o->interval != 0 && (o->seqCount % o->interval) == 0) return orc_flush(o);
    return ORC_OK;
}

/* batch form: reads concatenated in `bases`, read i = bases[offsets[i] .. offsets[i+1]) */
int orc_add_reads(orc_sketcher *o, const uint8_t *bases, const uint64_t *offsets, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) {
        int rc = orc_add_read(o, bases + offsets[i], (int32_t)(offsets[i + 1] - offsets[i]));
        if (rc != ORC_OK) return rc;
    }
    return ORC_OK;
}

/* final flush + StopWork + the "no sequences received" check (pipeline/sketch.go:219-239) */
int orc_finish(orc_sketcher *o) {
    int rc = orc_flush(o);
    if (rc != ORC_OK) return rc;
    if (o->seqCount == 0) return ORC_ERR_NOSEQ;
    return ORC_OK;
}

/* ---- accessors for tests */
void orc_get_sketch(const orc_sketcher *o, uint64_t *mins, double *weights) {
    memcpy(mins, o->sketch, sizeof(uint64_t) * o->S);
    memcpy(weights, o->weights, sizeof(double) * o->S);
}
void orc_get_histogram(const orc_sketcher *o, double *bins) { memcpy(bins, o->bins, sizeof(double) * (size_t)o->B); }
int32_t orc_used_bins(const orc_sketcher *o) { return o->used; }
int32_t orc_num_bins(const orc_sketcher *o) { return o->B; }
void orc_get_counters(const orc_sketcher *o, uint64_t *n_reads, uint64_t *n_minimizers,
                      uint64_t *total_len, uint64_t *n_flushes, uint64_t *n_elements) {
    *n_reads = o->seqCount; *n_minimizers = o->minimizerCounter; *total_len = o->lengthTotal;
    *n_flushes = o->flushes; *n_elements = o->elements;
}
void orc_get_cms(const orc_sketcher *o, double *ctr) {
    memcpy(ctr, o->cms.ctr, sizeof(double) * (size_t)o->cms.depth * o->cms.width);
}
const double *orc_cws_r(const orc_sketcher *o) { return o->r; }
const double *orc_cws_c(const orc_sketcher *o) { return o->c; }
const double *orc_cws_b(const orc_sketcher *o) { return o->b; }
/* histogram-only mode for tests that don't want to pay for the CWS tables:
 * add a histogram directly (as the collector would have built it) */
int orc_add_histogram(orc_sketcher *o, const uint32_t *hist) {
    for (int32_t i = 0; i < o->B; i++) {
        if (hist[i]) { if (o->bins[i] == 0.0) o->used++; o->bins[i] += (double)hist[i]; }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ smash (next-tier row)
 * distances.GetDistance "jaccard" — src/distances/distances.go:19-26;
 * distances.GetWJD — src/distances/distances.go:44-72;
 * HULKdata.GetDistance — src/sketchio/sketchio.go:259-306 (note :293-301: BOTH weight vectors of the
 * weighted Jaccard are taken from the SUBJECT sketch);  makeMatrix loop — cmd/smash.go:208-224. */
double orc_jaccard_distance(const uint64_t *a, const uint64_t *b, uint32_t n) {
    double intersect = 0.0;
    for (uint32_t i = 0; i < n; i++) if ((double)a[i] == (double)b[i]) intersect++;
    return 1.0 - (intersect / (double)n);
}
static double go_max(double x, double y) {            /* math.Max: NaN if either is NaN, +0 > -0 */
    if (isinf(x) && x > 0) return x;
    if (isinf(y) && y > 0) return y;
    if (x != x || y != y) return NAN;
    if (x == 0 && x == y) return signbit(x) ? y : x;
    return x > y ? x : y;
}
double orc_wjd(const uint64_t *setA, const uint64_t *setB, const double *weightsA, const double *weightsB, uint32_t n) {
    double intersect = 0.0, uni = 0.0;
    for (uint32_t i = 0; i < n; i++) {
        double weightA = go_max(go_max(weightsA[i], 0), go_max(-weightsA[i], 0));
        double weightB = go_max(go_max(weightsB[i], 0), go_max(-weightsB[i], 0));
        if ((double)setA[i] == (double)setB[i]) {
            if (weightA < weightB) { intersect += weightA; uni += weightB; }
            else { intersect += weightB; uni += weightA; }
        } else {
            if (weightA > weightB) uni += weightA; else uni += weightB;
        }
    }
    return 1 - (intersect / uni);
}
/* out[s*N + q] = distance(subject s, query q); metric 0 = jaccard, 1 = weightedjaccard */
void orc_smash_matrix(const uint64_t *mins, const double *weights, uint32_t N, uint32_t S, int metric, double *out) {
    for (uint32_t s = 0; s < N; s++)
        for (uint32_t q = 0; q < N; q++) {
            const uint64_t *a = mins + (size_t)s * S, *b = mins + (size_t)q * S;
            const double *wa = weights + (size_t)s * S;
            out[(size_t)s * N + q] = metric == 1 ? orc_wjd(a, b, wa, wa, S)      /* sketchio.go:296: hsB = subject */
                                                 : orc_jaccard_distance(a, b, S);
        }
}
