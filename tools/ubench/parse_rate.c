/* Host ingest rate without a GPU and without a consumer: hulk_parse_files with a callback that only counts.
 * build: gcc -O2 -o parse_rate parse_rate.c -I../../include -L../../hulk_amd/csrc -lhulkhip -Wl,-rpath,$PWD/../../hulk_amd/csrc
 * usage: parse_rate FILE [repeats [fasta]]      (HULK_GZ_PAR=0 / HULK_GZ_THREADS=n select the gzip reader; fasta = 1: --fasta mode) */
#include <stdio.h>
#include <stdlib.h>
#include "hulk_hip.h"
static int count(void *user, const uint8_t *bases, const uint64_t *offsets, uint64_t n) { *(uint64_t *)user += offsets[n] - offsets[0]; (void)bases; return 0; }
int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    const int fasta = argc > 3 ? atoi(argv[3]) : 0;
    const char *paths[1] = {argv[1]};
    for (int r = 0; r < reps; r++) {
        hulk_ingest_stats st; char err[256] = {0}; uint64_t bases = 0;
        const int rc = hulk_parse_files(paths, 1, fasta, 0, count, &bases, &st, err, sizeof err);
        if (rc) { printf("error %d: %s\n", rc, err); return 1; }
        printf("%llu reads, %llu bases, %.3f s: %.3g reads/s, %.2f GB/s of text\n", (unsigned long long)st.n_seqs, (unsigned long long)bases, st.seconds,
               st.n_seqs / st.seconds, st.bytes_in / st.seconds / 1e9);
    }
    return 0;
}
