// Micro-benchmark 2: what limits global atomicAdd throughput — lanes, cache lines, or returns?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
// MODE 0 random dword; 1 consecutive dwords per wave (random wave base); 2 one line per lane, consecutive lines;
// 3 random, returning; 4 random plain store (no atomic); 5 random plain load+store; 6 random, 16 active lanes only
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *hist, uint32_t nwords, int per_thread, uint32_t *sink) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t acc = 0;
    for (int i = 0; i < per_thread; i++) {
        uint32_t idx;
        const uint32_t r = (uint32_t)(mix(gid * 1000003ull + i) >> 20);
        const uint32_t rw = (uint32_t)(mix((gid >> 6) * 1000003ull + i) >> 20);
        if (MODE == 1) idx = ((rw % (nwords / 64)) * 64 + lane);
        else if (MODE == 2) idx = ((rw % (nwords / 2048)) * 2048 + lane * 32);
        else idx = r % nwords;
        if (MODE == 3) acc += atomicAdd(&hist[idx], 1u);
        else if (MODE == 4) hist[idx] = r;
        else if (MODE == 5) hist[idx] = hist[idx] + 1;
        else if (MODE == 6) { if (lane < 16) atomicAdd(&hist[idx], 1u); }
        else atomicAdd(&hist[idx], 1u);
    }
    if (acc == 0xdeadbeef) *sink = acc;
}
template <int MODE> int run(const char *name, uint32_t *d, uint32_t nwords, uint32_t *sink) {
    const int blocks = 4096, per = 26;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(d, nwords, per, sink);
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) k<MODE><<<blocks, 256>>>(d, nwords, per, sink);
    hipEventRecord(b); CHK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, a, b);
    const double n = 5.0 * blocks * 256 * per * (MODE == 6 ? 0.25 : 1.0);
    printf("%-46s %8.1f us per launch  %7.1f G lane-ops/s\n", name, ms * 1000 / 5, n / ms / 1e6);
    return 0;
}
int main() {
    const uint32_t nwords = 194481 / 64 * 64 * 1;   // ~778 KB
    uint32_t *d, *sink; CHK(hipMalloc(&d, (size_t)nwords * 4 * 16)); CHK(hipMalloc(&sink, 4)); CHK(hipMemset(d, 0, (size_t)nwords * 4 * 16));
    run<0>("random dwords, 778 KB", d, nwords, sink);
    run<0>("random dwords, 12 MB", d, nwords * 16, sink);
    run<1>("64 consecutive dwords per wave-op", d, nwords, sink);
    run<2>("one 128B line per lane, 64 adjacent lines", d, nwords, sink);
    run<3>("random dwords, returning", d, nwords, sink);
    run<4>("random plain 4B store", d, nwords, sink);
    run<5>("random plain load+store", d, nwords, sink);
    run<6>("random dwords, 16 of 64 lanes active", d, nwords, sink);
    return 0;
}
