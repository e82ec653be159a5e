// Single-core rates behind the .gz ingest figures: hulk::inflate vs zlib's inflate on FASTQ-like text, and zlib's crc32.
// build: g++ -O3 -std=c++17 -o inflate_rate inflate_rate.cpp -lz
#include "../../hulk_amd/csrc/fast_inflate.h"
#include <zlib.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace hulk::inflate;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = 256u << 20;
    std::vector<uint8_t> src(n);
    for (size_t i = 0, r = 0; i < n; r++) {
        i += (size_t)snprintf((char *)&src[i], n - i > 32 ? 32 : n - i, "@r%zu\n", r);
        for (int j = 0; j < 150 && i < n; j++) src[i++] = "ACGT"[rand() & 3];
        for (const char *p = "\n+\n"; *p && i < n; p++) src[i++] = (uint8_t)*p;
        for (int j = 0; j < 150 && i < n; j++) src[i++] = "FFFFFFF:,F"[rand() % 10];
        if (i < n) src[i++] = '\n';
    }
    for (int level : {1, 6}) {
        z_stream z{}; deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        std::vector<uint8_t> comp(deflateBound(&z, n) + 64);
        z.next_in = src.data(); z.avail_in = (uInt)n; z.next_out = comp.data(); z.avail_out = (uInt)comp.size();
        deflate(&z, Z_FINISH); const size_t cn = z.total_out; deflateEnd(&z);
        std::vector<uint8_t> win(n + 65536);
        Decoder *d = new Decoder; d->feed(comp.data(), cn);
        double t0 = now();
        uint8_t *out = win.data();
        while (d->state != Decoder::DONE && d->state != Decoder::ERROR) out = d->run(out, win.data() + win.size() - 400, win.data(), true);
        double dt = now() - t0;
        printf("level %d (ratio %.2f): hulk::inflate %.2f GB/s (%s)", level, (double)n / cn, n / dt / 1e9, memcmp(win.data(), src.data(), n) == 0 ? "equal" : "DIFFERENT");
        z_stream y{}; inflateInit2(&y, -15); y.next_in = comp.data(); y.avail_in = (uInt)cn; y.next_out = win.data(); y.avail_out = (uInt)n;
        t0 = now(); inflate(&y, Z_FINISH); dt = now() - t0; inflateEnd(&y);
        printf("   zlib inflate %.2f GB/s\n", n / dt / 1e9);
        delete d;
    }
    double t0 = now(); const uLong c = crc32(0, src.data(), (uInt)n); double dt = now() - t0;
    printf("zlib crc32 %.2f GB/s (%08lx)\n", n / dt / 1e9, c);
    return 0;
}
