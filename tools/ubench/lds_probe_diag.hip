// lds_probe_diag — the probe of hulk_countmin.hip (k_lds_order_probe) with every mismatching lane printed: which round, pattern,
// lane, addresses, the value returned and the value expected.  (Round 6: the in-library probe reported 32 lanes on its first run.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Rec { int r, l, p0, p1; unsigned long long au, ea, bu, eb; double ad, bd; };
__global__ __launch_bounds__(64) void k(Rec *out, int rounds) {
    __shared__ unsigned long long su[64];
    __shared__ double sd[64];
    const int l = threadIdx.x;
    uint32_t rng = 0x9E3779B9u * (uint32_t)(l + 1);
    for (int r = 0; r < rounds; r++) {
        const int mode = r & 3;
        rng = rng * 1664525u + 1013904223u; const uint32_t ra = rng >> 8;
        rng = rng * 1664525u + 1013904223u; const uint32_t rb = rng >> 8;
        const int p0 = mode == 0 ? 0 : mode == 1 ? l / 2 : (int)((ra * (uint32_t)(1 + r % 61)) >> 24);
        const int p1 = mode == 0 ? 0 : mode == 1 ? (63 - l) / 2 : (int)((rb * (uint32_t)(1 + r % 59)) >> 24);
        const unsigned long long v0 = 1ull + (unsigned)l, v1 = 1000ull + (unsigned)l;
        su[l] = 7ull * (unsigned)l; sd[l] = (double)(7 * l);
        __syncthreads();
        const unsigned long long a_u = atomicAdd(&su[p0], v0);
        const unsigned long long b_u = atomicAdd(&su[p1], v1);
        const double a_d = atomicAdd(&sd[p0], (double)v0);
        const double b_d = atomicAdd(&sd[p1], (double)v1);
        unsigned long long ea = 7ull * (unsigned)p0, eb = 7ull * (unsigned)p1;
        for (int j = 0; j < 64; j++) {
            const int q0 = __builtin_amdgcn_readlane(p0, j), q1 = __builtin_amdgcn_readlane(p1, j);
            if (q0 == p0 && j < l) ea += 1ull + (unsigned)j;
            if (q0 == p1) eb += 1ull + (unsigned)j;
            if (q1 == p1 && j < l) eb += 1000ull + (unsigned)j;
        }
        out[r * 64 + l] = Rec{r, l, p0, p1, a_u, ea, b_u, eb, a_d, b_d};
        __syncthreads();
    }
}
int main() {
    const int R = 488;
    Rec *d; hipMalloc((void **)&d, sizeof(Rec) * R * 64);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, R);
    std::vector<Rec> h(R * 64);
    hipMemcpy(h.data(), d, sizeof(Rec) * R * 64, hipMemcpyDeviceToHost);
    int bad = 0;
    for (auto &x : h) {
        const bool b = x.au != x.ea || x.bu != x.eb || x.ad != (double)x.ea || x.bd != (double)x.eb;
        if (b && bad++ < 80)
            printf("round %d mode %d lane %d p0 %d p1 %d: u64 first %llu (want %llu) second %llu (want %llu); f64 first %.0f second %.0f\n", x.r, x.r & 3, x.l, x.p0, x.p1,
                   x.au, x.ea, x.bu, x.eb, x.ad, x.bd);
    }
    printf("lds_probe_diag: %d of %d lanes mismatch\n", bad, R * 64);
    return 0;
}
