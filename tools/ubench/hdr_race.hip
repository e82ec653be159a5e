// hdr_race — does HIP keep the order "small fills + kernel on stream A  ->  event  ->  copy on stream B" on gfx950?
// Minimal reproducer for the anomaly hulk_step_sharded met in round 4 (docs/EXPERIMENTS.md: with GPU_MAX_HW_QUEUES=16 a
// device-to-host copy of a 128-byte exchange header delivered what the buffer held BEFORE the fills and kernels it was
// ordered behind).  R "ranks" = R host threads of one process, each with its own pair of non-blocking streams (A lowest
// priority like the flush stream, B highest like the exchange stream), optionally beside a thread that keeps the GPU busy.
// Variants (per iteration i, tag = i + 1; every variant checks all 32 words of the block it reads back):
//   1  A: memsetAsync(hdr, 0) ; memsetD32Async(hdr[0] = tag) ; k_write(hdr[1..31] = tag) ; eventRecord
//      B: streamWaitEvent ; memcpyAsync D2H (pinned) ; streamSynchronize(B)                    [the exchange's dependency]
//   2  B: memcpyAsync H2D (pinned, tag) ; streamSynchronize(B) ; eventRecord ; A: streamWaitEvent ; k_touch(read) ;
//      memcpyAsync D2H ; eventRecord ; eventSynchronize                                        [the header copy after an exchange]
//   3  everything of variant 1 on stream A alone                                               [one stream]
//   4  variant 1 with hipStreamSynchronize(A) instead of the event                             [host-ordered, two streams]
//   5  variant 1, the read-back by a kernel that stores to mapped pinned memory instead of a copy
//   6  variant 1 into a FRESH pinned buffer: hipHostMalloc right in front of the copy, hipHostFree after the check
//      (the host transport's staging of step 0 was allocated like that)
//   7  variant 1 on a rank with FOUR streams (two more that carry a small kernel each iteration): 4 x ranks streams in all
// usage: hdr_race [iterations] [ranks] [load 0/1] [variants, e.g. 1234567]     (set GPU_MAX_HW_QUEUES in the environment)
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void k_write(uint32_t *hdr, uint32_t tag) { if (threadIdx.x >= 1 && threadIdx.x < 32) hdr[threadIdx.x] = tag; }
__global__ void k_touch(const uint32_t *buf, uint32_t *sink, int words) {
    uint32_t a = 0;
    for (int i = threadIdx.x; i < words; i += blockDim.x) a += buf[i];
    if (a == 0xdeadbeef) *sink = a;
}
__global__ void k_copy_out(const uint32_t *hdr, volatile uint32_t *out) {
    if (threadIdx.x < 32) out[threadIdx.x] = hdr[threadIdx.x];
    __threadfence_system();
}
__global__ void k_load(float *p, int n, int rounds) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = i < n ? p[i] : 0.f;
    for (int r = 0; r < rounds; r++) v = v * 1.0001f + 0.5f;
    if (i < n) p[i] = v;
}

struct Tally { std::atomic<unsigned long long> iters{0}, stale_fill{0}, stale_kernel{0}, torn{0}; };

static void rank_main(int rank, int iters, const char *variants, Tally *tally) {
    CHK(hipSetDevice(0));
    int lo = 0, hi = 0;
    CHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t A, B, C, D;
    CHK(hipStreamCreateWithPriority(&A, hipStreamNonBlocking, lo));
    CHK(hipStreamCreateWithPriority(&B, hipStreamNonBlocking, hi));
    CHK(hipStreamCreateWithFlags(&C, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&D, hipStreamNonBlocking));
    hipEvent_t ev, ev2;
    CHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CHK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
    uint32_t *d_hdr, *d_sink, *h_dst, *h_src;
    CHK(hipMalloc((void **)&d_hdr, 32 * 4 * 8));
    CHK(hipMalloc((void **)&d_sink, 4));
    CHK(hipMemset(d_hdr, 0, 32 * 4 * 8));
    CHK(hipDeviceSynchronize());
    CHK(hipHostMalloc((void **)&h_dst, 32 * 4, hipHostMallocDefault));
    CHK(hipHostMalloc((void **)&h_src, 32 * 4, hipHostMallocDefault));
    for (const char *v = variants; *v; v++) {
        Tally &t = tally[*v - '0'];
        for (int i = 0; i < iters; i++) {
            const uint32_t tag = (uint32_t)i + 1;
            memset(h_dst, 0xee, 128);
            switch (*v) {
                case '1': case '3': case '4': case '5': case '6': case '7': {
                    hipStream_t rd = (*v == '3') ? A : B;
                    uint32_t *dst = h_dst;
                    if (*v == '6') { CHK(hipHostMalloc((void **)&dst, 4096, hipHostMallocDefault)); memset(dst, 0xee, 128); }
                    if (*v == '7') { hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, C, d_hdr + 64, d_sink, 32); hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, D, d_hdr + 64, d_sink, 32); }
                    CHK(hipMemsetAsync(d_hdr, 0, 128, A));
                    CHK(hipMemsetD32Async((hipDeviceptr_t)d_hdr, (int)tag, 1, A));
                    hipLaunchKernelGGL(k_write, dim3(1), dim3(64), 0, A, d_hdr, tag);
                    if (*v == '4') CHK(hipStreamSynchronize(A));
                    else if (*v != '3') { CHK(hipEventRecord(ev, A)); CHK(hipStreamWaitEvent(B, ev, 0)); }
                    if (*v == '5') hipLaunchKernelGGL(k_copy_out, dim3(1), dim3(64), 0, rd, d_hdr, h_dst);
                    else CHK(hipMemcpyAsync(dst, d_hdr, 128, hipMemcpyDeviceToHost, rd));
                    CHK(hipStreamSynchronize(rd));
                    if (*v == '6') { memcpy(h_dst, dst, 128); CHK(hipHostFree(dst)); }
                    break;
                }
                case '2': {
                    for (int j = 0; j < 32; j++) h_src[j] = tag;
                    CHK(hipMemcpyAsync(d_hdr, h_src, 128, hipMemcpyHostToDevice, B));
                    CHK(hipStreamSynchronize(B));
                    CHK(hipEventRecord(ev, B));
                    CHK(hipStreamWaitEvent(A, ev, 0));
                    hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, A, d_hdr, d_sink, 32);
                    CHK(hipMemcpyAsync(h_dst, d_hdr, 128, hipMemcpyDeviceToHost, A));
                    CHK(hipEventRecord(ev2, A));
                    CHK(hipEventSynchronize(ev2));
                    break;
                }
                default: break;
            }
            const bool fill_ok = h_dst[0] == tag;
            int kernel_bad = 0;
            for (int j = 1; j < 32; j++) kernel_bad += h_dst[j] != tag;
            t.iters++;
            if (!fill_ok) t.stale_fill++;
            if (kernel_bad) t.stale_kernel++;
            if ((!fill_ok) != (kernel_bad == 31) && (!fill_ok || kernel_bad)) t.torn++;
            if ((!fill_ok || kernel_bad) && t.stale_fill + t.stale_kernel <= 5)
                fprintf(stderr, "variant %c rank %d iter %d: word0 %u (want %u), %d of 31 kernel words wrong (first %u)\n", *v, rank, i,
                        h_dst[0], tag, kernel_bad, h_dst[1]);
        }
    }
    CHK(hipStreamSynchronize(A)); CHK(hipStreamSynchronize(B)); CHK(hipStreamSynchronize(C)); CHK(hipStreamSynchronize(D));
    hipStreamDestroy(C); hipStreamDestroy(D);
    hipFree(d_hdr); hipFree(d_sink); hipHostFree(h_dst); hipHostFree(h_src);
    hipEventDestroy(ev); hipEventDestroy(ev2); hipStreamDestroy(A); hipStreamDestroy(B);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 100000;
    const int ranks = argc > 2 ? atoi(argv[2]) : 4;
    const int load = argc > 3 ? atoi(argv[3]) : 0;
    const char *variants = argc > 4 ? argv[4] : "12345";
    CHK(hipSetDevice(0));
    Tally tally[10];
    std::atomic<bool> stop{false};
    std::thread loader;
    if (load) loader = std::thread([&] {
        CHK(hipSetDevice(0));
        hipStream_t s[2]; float *p;
        const int n = 64 << 20;
        CHK(hipMalloc((void **)&p, (size_t)n * 4));
        for (auto &x : s) CHK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
        int i = 0;
        while (!stop) { hipLaunchKernelGGL(k_load, dim3(n / 256), dim3(256), 0, s[i & 1], p, n, 200); if ((++i & 7) == 0) CHK(hipStreamSynchronize(s[i & 1])); }
        for (auto &x : s) { CHK(hipStreamSynchronize(x)); hipStreamDestroy(x); }
        hipFree(p);
    });
    std::vector<std::thread> th;
    for (int r = 0; r < ranks; r++) th.emplace_back(rank_main, r, iters, variants, tally);
    for (auto &t : th) t.join();
    stop = true;
    if (load) loader.join();
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    for (const char *v = variants; *v; v++) {
        Tally &t = tally[*v - '0'];
        printf("GPU_MAX_HW_QUEUES=%s ranks=%d load=%d variant %c: %llu iterations, stale fill word %llu, stale kernel words %llu, torn blocks %llu\n",
               q ? q : "default", ranks, load, *v, (unsigned long long)t.iters, (unsigned long long)t.stale_fill,
               (unsigned long long)t.stale_kernel, (unsigned long long)t.torn);
    }
    return 0;
}
