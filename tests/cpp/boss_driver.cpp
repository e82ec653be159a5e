// Drives the C++ host mirror (include/hulk.hpp) the way SeqMinimizer.Run / Sketcher.Run drive the Go
// objects; prints one JSON line that tests/test_gpu_cpp_host.py compares with the Python/ctypes path.
//   boss_driver addseq <reads.txt> k w S interval decay      one sequence per line -> AddSeq
//   boss_driver files  <path>      k w S interval decay      SketchFiles (native ingest)
//   boss_driver sharded <reads.txt> k w S interval decay     the same stream through Shard (RCCL, world size 1) + AddSeq +
//                                                             StopWorkSharded: hulk_step_sharded_host per full share
//   boss_driver errors                                         the reference's fatal messages
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include "hulk.hpp"

static void print_sketch(hulk::Boss &boss, uint64_t seqs) {
    const hulk::HistoSketch hs = boss.Sketch();
    std::printf("{\"n_seqs\": %llu, \"n_minimizers\": %llu, \"ksize\": %u, \"num\": %u, \"bins\": %d, \"drift\": %s, \"mins\": [",
                (unsigned long long)seqs, (unsigned long long)boss.GetMinimizerCount(), hs.KmerSize, hs.SketchSize,
                hs.Dimensions, hs.ApplyConceptDrift ? "true" : "false");
    for (size_t i = 0; i < hs.Sketch.size(); i++) std::printf("%s%llu", i ? ", " : "", (unsigned long long)hs.Sketch[i]);
    std::printf("], \"weights\": [");
    for (size_t i = 0; i < hs.SketchWeights.size(); i++) std::printf("%s%.17g", i ? ", " : "", hs.SketchWeights[i]);
    std::printf("]}\n");
}

int main(int argc, char **argv) {
    const std::string mode = argc > 1 ? argv[1] : "";
    try {
        if (mode == "errors") {
            // NewMinimizerSketch / NewHistoSketch checks, same texts as the reference
            const struct { unsigned k, w, s; double x; } bad[] = {{21, 300, 8, 1.0}, {40, 9, 8, 1.0}, {21, 9, 8, 1.5}};
            for (const auto &b : bad) {
                hulk::SketchInfo info; info.KmerSize = b.k; info.WindowSize = b.w; info.SketchSize = b.s; info.DecayRatio = b.x;
                try { hulk::Boss::FindMinimizers(info); std::printf("no error\n"); }
                catch (const hulk::Error &e) { std::printf("%d|%s\n", e.code(), e.what()); }
            }
            hulk::SketchInfo info; info.SketchSize = 8;
            hulk::Boss boss = hulk::Boss::FindMinimizers(info);
            try { boss.AddSeq("ACGTACGT"); boss.StopWork(); std::printf("no error\n"); }
            catch (const hulk::Error &e) { std::printf("%d|%s\n", e.code(), e.what()); }
            return 0;
        }
        if (mode == "smashfiles") {
            // boss_driver smashfiles <out.csv> <metric> <k> <file.json>...: `hulk smash` through hulk::SmashFiles
            if (argc < 7) { std::fprintf(stderr, "usage: boss_driver smashfiles <out.csv> <metric> <k> <a.json> <b.json> ...\n"); return 2; }
            std::vector<std::string> files(argv + 5, argv + argc);
            hulk::SmashStats st;
            const std::vector<double> d = hulk::SmashFiles(files, (uint32_t)std::atoi(argv[4]), "histosketch", argv[3], argv[2], std::string(), &st);
            std::printf("{\"n\": %u, \"size\": %u, \"d01\": %.17g}\n", st.Sketches, st.SketchSize, d.size() > 1 ? d[1] : -1.0);
            return 0;
        }
        if (argc < 8) { std::fprintf(stderr, "usage: boss_driver addseq|files|sharded <path> k w S interval decay\n"); return 2; }
        hulk::SketchInfo info;
        info.KmerSize = (unsigned)std::atoi(argv[3]); info.WindowSize = (unsigned)std::atoi(argv[4]);
        info.SketchSize = (unsigned)std::atoi(argv[5]); info.Interval = (unsigned)std::atoi(argv[6]);
        info.DecayRatio = std::atof(argv[7]);
        hulk::Boss theBoss = hulk::Boss::FindMinimizers(info);
        uint64_t seqCount = 0;
        if (mode == "sharded") {
            theBoss.Shard(hulk::Boss::CommUniqueId());                                // one rank: the id needs no other host
            std::ifstream in(argv[2]);
            std::string line;
            while (std::getline(in, line)) { theBoss.AddSeq(line); seqCount++; }
            const uint64_t perStep = (uint64_t)hulk_batch_size(theBoss.handle()) * info.Interval;
            const uint64_t rest = seqCount % perStep;
            theBoss.StopWorkSharded((uint32_t)((rest + info.Interval - 1) / info.Interval));
            print_sketch(theBoss, seqCount);
            return 0;
        }
        if (mode == "addseq") {
            std::ifstream in(argv[2]);
            std::string line;
            while (std::getline(in, line)) { theBoss.AddSeq(line); seqCount++; }     // sketch.go:196-217
        } else {
            seqCount = theBoss.SketchFiles({argv[2]}).SeqCount;
        }
        theBoss.StopWork();                                                           // sketch.go:219-224
        print_sketch(theBoss, seqCount);
    } catch (const hulk::Error &e) {
        std::printf("ERROR---> %s\n", e.what());
        return 1;
    }
    return 0;
}
