// hulk_tables.hip — the static tables of a context: count-min chain tables and the CWS parameter matrices r, c, b
// (newCWS, src/histosketch/histosketch.go:95-126; go_rng over Go's math/rand, seed 1).  Host orchestration; the arithmetic
// runs in hulk_countmin.hip (k_build_chains) and hulk_cws.hip (k_alfg_*, k_cws_eval / scatter / beta, k_build_k32).
#include "hulk_ctx.h"
#include "cws_gen.h"
#include "go_rng_jump.h"

#include <algorithm>
#include <cmath>

namespace hulk {
namespace {
// go-jump on the host, only for building the static count-min chain tables
int32_t jump_host(uint64_t key, int64_t n) {
    int64_t b = -1, j = 0;
    if (n <= 0) n = 1;
    while (j < n) {
        b = j;
        key = key * 2862933555777941757ull + 1;
        j = (int64_t)((double)(b + 1) * ((double)(1LL << 31) / (double)((key >> 33) + 1)));
    }
    return (int32_t)b;
}

}  // namespace

// static chain tables for the count-min prefix sums: for row d, bins grouped by counter
// position g = jump(bin*(d+1), width) (countmin.go:122-125), ascending bin inside a group.
int build_chains(hulk_ctx *c) {
    const int D = c->cms_depth, W = c->cms_width; const int32_t B = c->B;
    if (!HULK_EXP_ENV("HULK_CHAINS_HOST")) {          // (the host loop below is kept as the A/B check of k_build_chains)
        HIPCHK(c, dalloc(&c->d_meta8, (size_t)D * B));
        HIPCHK(c, dalloc(&c->d_pos16, (size_t)D * B));
        HIPCHK(c, launch_build_chains(c->stream, c->d_pos16, c->d_meta8, B, D, W));
        return HULK_OK;
    }
    std::vector<uint32_t> pos(B);
    std::vector<uint16_t> pos16((size_t)D * B);
    std::vector<uint8_t> meta8((size_t)D * B);       // bits 0-6: previous lane of the 64-bin chunk on the same counter (64 = none); bit 7: last one
    for (int d = 0; d < D; d++) {
        for (int32_t b = 0; b < B; b++) {
            uint64_t h = (uint64_t)b + (uint64_t)d * (uint64_t)b;      // countmin.go:123-125: hash(bin + d * bin)
            pos[b] = (uint32_t)jump_host(h, W);
            pos16[(size_t)d * B + b] = (uint16_t)pos[b];
        }
        std::vector<int32_t> last_bin(W, -1);
        for (int32_t c0 = 0; c0 < B; c0 += 64) {
            const int32_t c1 = std::min<int32_t>(B, c0 + 64);
            for (int32_t b = c0; b < c1; b++) {
                const int32_t prev = last_bin[pos[b]];
                uint8_t m = 64;
                if (prev >= c0) { m = (uint8_t)(prev - c0); meta8[(size_t)d * B + prev] &= 0x7f; }   // prev is no longer the last
                meta8[(size_t)d * B + b] = m | 0x80;
                last_bin[pos[b]] = b;
            }
        }
    }
    HIPCHK(c, dalloc(&c->d_meta8, meta8.size()));
    HIPCHK(c, hipMemcpy(c->d_meta8, meta8.data(), meta8.size(), hipMemcpyHostToDevice));
    HIPCHK(c, dalloc(&c->d_pos16, pos16.size()));
    HIPCHK(c, hipMemcpy(c->d_pos16, pos16.data(), pos16.size() * 2, hipMemcpyHostToDevice));
    return HULK_OK;
}

// upload r,c,b rows owned by this context as interleaved {r,c,b} and derive the fp32 K table
int install_tables(hulk_ctx *c, const double *r, const double *cc, const double *b) {
    const size_t B = (size_t)c->B;
    std::vector<double> row(B * 3);
    for (uint32_t s = 0; s < c->slots; s++) {
        const size_t src = (size_t)(c->slot_begin + s) * B;
        for (size_t j = 0; j < B; j++) { row[j * 3] = r[src + j]; row[j * 3 + 1] = cc[src + j]; row[j * 3 + 2] = b[src + j]; }
        HIPCHK(c, hipMemcpy(c->d_rcb + (size_t)s * B * 3, row.data(), B * 3 * sizeof(double), hipMemcpyHostToDevice));
    }
    HIPCHK(c, launch_build_k32(c->stream, c->d_rcb, c->d_k32, (int)c->slots, c->B, c->row_stride));
    if (c->slots) {
        HIPCHK(c, launch_tile_kmin(c->stream, c->d_k32, c->d_kmin32, (int)c->slots, c->ntiles, c->row_stride));
        HIPCHK(c, launch_slot_kmin(c->stream, c->d_kmin32, c->d_kminslot, (int)c->slots, c->ntiles));
    }
    c->tables_ready = true;
    return HULK_OK;
}

// newCWS (histosketch.go:95-126): the host walks Go's math/rand streams (cws_gen.h), the device does
// the gamma math and the in-order compaction, chunk by chunk (double-buffered pinned staging).
int generate_tables_host(hulk_ctx *c) {
    const uint64_t B = (uint64_t)c->B;
    const uint64_t need_entries = (uint64_t)(c->slot_begin + c->slots) * B;     // rows of later slots are not needed
    if (need_entries == 0) { c->tables_ready = true; return HULK_OK; }
    const uint64_t need_gammas = 2 * need_entries;
    const size_t CH = (size_t)1 << 22;                                        // attempts (or uniforms) per chunk
    const CwsConstants K((c->p.flags & HULK_FLAG_GAMMA_CPYTHON) != 0);
    uint64_t *h_buf[2] = {nullptr, nullptr}; uint64_t *d_pairs[2] = {nullptr, nullptr};
    double *d_val = nullptr; uint32_t *d_blkcnt = nullptr; unsigned long long *d_tot = nullptr;
    hipEvent_t done[2] = {nullptr, nullptr};
    int rc = HULK_OK;
    auto cleanup = [&]() {
        for (int i = 0; i < 2; i++) { if (h_buf[i]) hipHostFree(h_buf[i]); hipFree(d_pairs[i]); if (done[i]) hipEventDestroy(done[i]); }
        hipFree(d_val); hipFree(d_blkcnt); hipFree(d_tot);
    };
#define GEN_CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = fail_hip(c, e_, #call); cleanup(); return rc; } } while (0)
    for (int i = 0; i < 2; i++) {
        GEN_CHK(hipHostMalloc((void **)&h_buf[i], CH * 16, hipHostMallocDefault));
        GEN_CHK(hipMalloc((void **)&d_pairs[i], CH * 16));
        GEN_CHK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
    }
    GEN_CHK(hipMalloc((void **)&d_val, CH * 8));
    GEN_CHK(hipMalloc((void **)&d_blkcnt, (CH / 1024 + 1) * 4));
    GEN_CHK(hipMalloc((void **)&d_tot, 16));
    GEN_CHK(hipMemsetAsync(d_tot, 0, 16, c->stream));
    // ---- r and c: gamma variates
    {
        AttemptStream attempts;
        unsigned long long got = 0; int cur = 0; uint64_t inflight[2] = {0, 0};
        // acceptance of Cheng's sampler at alpha = 2 is ~0.8; the tail chunk is sized from the estimate
        while (got < need_gammas) {
            uint64_t want = (uint64_t)((double)(need_gammas - got) / 0.78) + 4096;
            if (want > CH) want = CH;
            if (inflight[cur]) GEN_CHK(hipEventSynchronize(done[cur]));         // staging buffer free again
            attempts.fill(h_buf[cur], (size_t)want);
            GEN_CHK(hipMemcpyAsync(d_pairs[cur], h_buf[cur], want * 16, hipMemcpyHostToDevice, c->stream));
            GEN_CHK(launch_cws_chunk(c->stream, d_pairs[cur], want, d_val, d_blkcnt, d_tot, d_tot + 1, c->d_rcb, B,
                                     c->slot_begin, c->slots, c->S, K.ainv, K.bbb, K.ccc, K.magic, nullptr, 0, nullptr, 0));
            GEN_CHK(hipEventRecord(done[cur], c->stream));
            inflight[cur] = want;
            cur ^= 1;
            // progress is only needed near the end; until then overlap host generation with the device
            if ((double)(got + (unsigned long long)(0.70 * (double)want)) >= (double)need_gammas || want < CH) {
                GEN_CHK(hipMemcpyAsync(&got, d_tot, 8, hipMemcpyDeviceToHost, c->stream));
                GEN_CHK(hipStreamSynchronize(c->stream));
            } else {
                got += (unsigned long long)(0.70 * (double)want);                // safe under-estimate
            }
        }
    }
    // ---- b = U(0,1) * r
    {
        UniformStream uni;
        int cur = 0; bool used[2] = {false, false};
        for (uint64_t first = 0; first < need_entries; first += CH) {
            const uint64_t n = std::min<uint64_t>(CH, need_entries - first);
            if (used[cur]) GEN_CHK(hipEventSynchronize(done[cur]));
            uni.fill(h_buf[cur], (size_t)n);
            GEN_CHK(hipMemcpyAsync(d_pairs[cur], h_buf[cur], n * 8, hipMemcpyHostToDevice, c->stream));
            GEN_CHK(launch_cws_beta(c->stream, d_pairs[cur], first, n, c->d_rcb, B, c->slot_begin, c->slots));
            GEN_CHK(hipEventRecord(done[cur], c->stream));
            used[cur] = true; cur ^= 1;
        }
    }
    GEN_CHK(launch_build_k32(c->stream, c->d_rcb, c->d_k32, (int)c->slots, c->B, c->row_stride));
    if (c->slots) {
        GEN_CHK(launch_tile_kmin(c->stream, c->d_k32, c->d_kmin32, (int)c->slots, c->ntiles, c->row_stride));
        GEN_CHK(launch_slot_kmin(c->stream, c->d_kmin32, c->d_kminslot, (int)c->slots, c->ntiles));
    }
    GEN_CHK(hipStreamSynchronize(c->stream));
#undef GEN_CHK
    cleanup();
    c->tables_ready = true;
    return HULK_OK;
}

// The same tables with the math/rand stream generated ON THE DEVICE (k_alfg_jump / k_alfg_fill: chunks of 2^20
// values started in parallel through the jump polynomial of go_rng_jump.h).  Both go_rng generators are seeded with
// 1, so ONE raw stream serves the gamma attempts (two values each) and the uniforms (one per entry).  The only
// data-dependent part of the consumption — an attempt whose u1 fails the range test takes one value instead of two,
// 2e-7 of them — is found by k_rng_candidates and resolved here into the `ev` list k_cws_eval uses; a value that
// would make Float64() resample (2^-54) sends the whole generation to the host walk instead.
// returns HULK_OK, an error, or +1 = "use the host generator".
int generate_tables_device(hulk_ctx *c) {
    const uint64_t B = (uint64_t)c->B;
    const uint64_t need_entries = (uint64_t)(c->slot_begin + c->slots) * B;
    if (need_entries == 0) { c->tables_ready = true; return HULK_OK; }
    const uint64_t need_gammas = 2 * need_entries;
    const uint64_t C = 1ull << GO_RNG_JUMP_LOG2;
    const size_t CH = (size_t)1 << 22;                                        // attempts per evaluation chunk
    const CwsConstants K((c->p.flags & HULK_FLAG_GAMMA_CPYTHON) != 0);
    // Cheng's sampler accepts ~77 % of the attempts at alpha = 2; the stream is sized with a wide margin
    const uint64_t max_attempts = (uint64_t)((double)need_gammas / 0.66) + (1u << 20);
    const uint64_t n_chunks = (2 * max_attempts + 4096 + C - 1) / C;
    const uint64_t total_raw = n_chunks * C;
    uint64_t *d_raw = nullptr, *d_win = nullptr, *d_coef = nullptr, *d_list = nullptr, *d_ev = nullptr;
    unsigned int *d_cnt = nullptr; double *d_val = nullptr; uint32_t *d_blkcnt = nullptr; unsigned long long *d_tot = nullptr;
    const uint32_t LIST_CAP = 1u << 16;
    int rc = HULK_OK;
    auto cleanup = [&]() { hipFree(d_raw); hipFree(d_win); hipFree(d_coef); hipFree(d_list); hipFree(d_ev); hipFree(d_cnt);
                           hipFree(d_val); hipFree(d_blkcnt); hipFree(d_tot); };
#define GEN_CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = fail_hip(c, e_, #call); cleanup(); return rc; } } while (0)
    if (hipMalloc((void **)&d_raw, total_raw * 8) != hipSuccess) { (void)hipGetLastError(); cleanup(); return 1; }   // not enough HBM: host walk
    GEN_CHK(hipMalloc((void **)&d_win, n_chunks * 607 * 8));
    GEN_CHK(hipMalloc((void **)&d_coef, 2 * 607 * 8));                         // x^(2^20) and x^(2^26)
    GEN_CHK(hipMalloc((void **)&d_list, (size_t)LIST_CAP * 8));
    GEN_CHK(hipMalloc((void **)&d_cnt, 4));
    GEN_CHK(hipMalloc((void **)&d_val, CH * 8));
    GEN_CHK(hipMalloc((void **)&d_blkcnt, (CH / 1024 + 1) * 4));
    GEN_CHK(hipMalloc((void **)&d_tot, 16));
    {
        uint64_t w0[607];
        GoRandSource(1).initial_window(w0);
        GEN_CHK(hipMemcpyAsync(d_win, w0, sizeof w0, hipMemcpyHostToDevice, c->stream));
        GEN_CHK(hipMemcpyAsync(d_coef, GO_RNG_JUMP, 607 * 8, hipMemcpyHostToDevice, c->stream));
        GEN_CHK(hipMemcpyAsync(d_coef + 607, GO_RNG_JUMP_FAR, 607 * 8, hipMemcpyHostToDevice, c->stream));
        GEN_CHK(hipMemsetAsync(d_cnt, 0, 4, c->stream));
        GEN_CHK(hipMemsetAsync(d_tot, 0, 16, c->stream));
        GEN_CHK(hipStreamSynchronize(c->stream));                              // w0 is a stack buffer
    }
    static const bool one_level = HULK_EXP_ENV("HULK_ALFG_ONE_LEVEL") != nullptr;      // A/B aid: the single walk over all chunks
    GEN_CHK(launch_alfg(c->stream, d_coef, one_level ? nullptr : d_coef + 607, 1u << (GO_RNG_JUMP_FAR_LOG2 - GO_RNG_JUMP_LOG2),
                        d_win, d_raw, 0, (uint32_t)n_chunks, C));
    GEN_CHK(launch_rng_candidates(c->stream, d_raw, total_raw, d_list, LIST_CAP, d_cnt));
    unsigned int n_cand = 0;
    GEN_CHK(hipMemcpyAsync(&n_cand, d_cnt, 4, hipMemcpyDeviceToHost, c->stream));
    GEN_CHK(hipStreamSynchronize(c->stream));
    if (n_cand > LIST_CAP) { cleanup(); return 1; }
    std::vector<uint64_t> cand(n_cand), ev;
    if (n_cand) GEN_CHK(hipMemcpy(cand.data(), d_list, (size_t)n_cand * 8, hipMemcpyDeviceToHost));
    std::sort(cand.begin(), cand.end());
    {   // walk the candidates: p0 = stream position of the u1 of valid attempt i0
        uint64_t p0 = 0, i0 = 0;
        for (uint64_t cd : cand) {
            const uint64_t pos = cd >> 1;
            if (cd & 1) { cleanup(); return 1; }                               // Float64() would resample here: host walk
            if (pos < p0 || ((pos - p0) & 1)) continue;                        // a u2 position: no range test there
            const uint64_t k = i0 + (pos - p0) / 2;                            // the valid attempt that follows the dead one
            ev.push_back(k);
            p0 = pos + 1; i0 = k;
        }
    }
    if (!ev.empty()) {
        GEN_CHK(hipMalloc((void **)&d_ev, ev.size() * 8));
        GEN_CHK(hipMemcpy(d_ev, ev.data(), ev.size() * 8, hipMemcpyHostToDevice));
    }
    // ---- r and c: gamma variates, in chunks of CH attempts straight from the device stream
    {
        unsigned long long got = 0; uint64_t next_attempt = 0;
        while (got < need_gammas) {
            uint64_t plan = (uint64_t)((double)(need_gammas - got) / 0.80) + 4096;   // a slight under-estimate: no overshoot of chunks
            while (plan > 0) {
                const uint64_t want = std::min<uint64_t>(plan, CH);
                if (2 * (next_attempt + want) + ev.size() + 2 > total_raw) { cleanup(); return 1; }   // margin exhausted (not expected)
                GEN_CHK(launch_cws_chunk(c->stream, nullptr, want, d_val, d_blkcnt, d_tot, d_tot + 1, c->d_rcb, B,
                                         c->slot_begin, c->slots, c->S, K.ainv, K.bbb, K.ccc, K.magic, d_raw, next_attempt,
                                         d_ev, (uint32_t)ev.size()));
                next_attempt += want; plan -= want;
            }
            GEN_CHK(hipMemcpyAsync(&got, d_tot, 8, hipMemcpyDeviceToHost, c->stream));
            GEN_CHK(hipStreamSynchronize(c->stream));
        }
    }
    // ---- b = U(0,1) * r: entry e takes stream value e (the uniform generator is a second source with the same seed)
    GEN_CHK(launch_cws_beta(c->stream, d_raw, 0, need_entries, c->d_rcb, B, c->slot_begin, c->slots));
    GEN_CHK(launch_build_k32(c->stream, c->d_rcb, c->d_k32, (int)c->slots, c->B, c->row_stride));
    if (c->slots) {
        GEN_CHK(launch_tile_kmin(c->stream, c->d_k32, c->d_kmin32, (int)c->slots, c->ntiles, c->row_stride));
        GEN_CHK(launch_slot_kmin(c->stream, c->d_kmin32, c->d_kminslot, (int)c->slots, c->ntiles));
    }
    GEN_CHK(hipStreamSynchronize(c->stream));
#undef GEN_CHK
    cleanup();
    c->tables_ready = true;
    return HULK_OK;
}

int generate_tables(hulk_ctx *c) {
    static const bool host_only = HULK_EXP_ENV("HULK_CWS_HOST") != nullptr;
    if (!host_only) {
        const int rc = generate_tables_device(c);
        if (rc <= 0) return rc;                                                // done, or a real error
    }
    return generate_tables_host(c);
}

int ensure_tables(hulk_ctx *c) {
    if (c->tables_ready) return HULK_OK;
    if (c->p.cws_source == HULK_CWS_EXTERNAL)
        return fail(c, HULK_ERR_STATE, "cws_source is EXTERNAL but hulk_set_cws_tables was not called");
    return generate_tables(c);
}

}  // namespace hulk
