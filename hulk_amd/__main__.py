"""`python -m hulk_amd sketch|smash ...` — the `hulk sketch` / `hulk smash` flag surface (cmd/root.go:62-66,
cmd/sketch.go:50-59, cmd/smash.go) over the GPU path, writing the reference's JSON sketch / similarity matrix.

Flag handling, log lines and the writers only: the line pump (src/pipeline/sketch.go:40-161) and
every numeric step run in libhulkhip.  --khf/--kmv (never fed in the reference) and --profiling are not provided.
"""
import argparse
import os
import sys
import time

from . import GpuSketcher, HulkError, spectrum_size
from .sketchio import HULKdata, VERSION


def log(msg):
    print(time.strftime("%Y/%m/%d %H:%M:%S ") + msg, flush=True)


def run_sketch(a):
    start = time.time()
    log(f"this is hulk (version {VERSION})")
    log("please cite Rowe et al. 2019, doi: https://doi.org/10.1186/s40168-019-0653-2")
    log("starting the sketch subcommand")
    log("checking parameters...")
    for f in a.fastq:
        if not os.path.exists(f):
            raise HulkError(-30, f"file does not exist: {f}")
    log("\tmode: FASTA" if a.fasta else "\tmode: FASTQ")
    log(f"\tminimizer k-mer size: {a.kmerSize}")
    log(f"\tminimizer window size: {a.windowSize}")
    log(f"\tsketch size: {a.sketchSize}")
    if a.decayRatio == 1:
        log("\tconcept drift: disabled")
    else:
        log("\tconcept drift: enabled")
        log(f"\tdecay ratio: {a.decayRatio:.2f}")
    bins = spectrum_size(a.kmerSize)
    log(f"\tnumber of bins in k-mer spectrum: {bins}")
    log("initialising sketching pipeline...")
    g = GpuSketcher(a.kmerSize, a.windowSize, a.sketchSize, a.interval, a.decayRatio, device=a.device)
    log("finding minimizers...")
    # DataStreamer + FastqHandler + the AddSeq loop run in libhulkhip (hulk_sketch_files); the
    # per-100k progress lines of sketch.go:204-207 are printed once the input has been consumed
    st = g.sketch_files(a.fastq, fasta=a.fasta, threads=a.processors if a.processors > 1 else 0)
    seq_count, length_total = st["n_seqs"], st["total_len"]
    for done in range(100000, seq_count + 1, 100000):
        log(f"\tprocessed {done} sequences")
    log("generating final histosketch of k-mer spectra...")
    if seq_count == 0:
        raise HulkError(-10, "no sequences received")
    g.stop_work()
    log(f"\tprocessed {seq_count} sequences in total")
    log(f"\tmean sequence length: {int(length_total / seq_count)}")
    log(f"\tfound {g.get_minimizer_count()} minimizers")
    log(f"\thistosketching across {bins} bins")
    log("cleaning up...")
    d = HULKdata()
    d.add(g.histosketch())
    d.filename = "".join(f + "," for f in a.fastq) if a.fastq else "STDIN"   # cmd/sketch.go:147-155
    d.banner_label = a.bannerLabel
    out = a.outFile + ".json"
    od = os.path.dirname(a.outFile)
    if od and od != "." and not os.path.exists(od):
        os.makedirs(od, mode=0o700)
    d.write_json(out)
    log(f"\twritten sketch to disk: {out}")
    g.close()
    log(f"finished in {time.time() - start:.3f}s")


def main(argv=None):
    ap = argparse.ArgumentParser(prog="hulk", description="Histosketching Using Little Kmers (MI355X path)")
    sub = ap.add_subparsers(dest="cmd", required=True)
    sk = sub.add_parser("sketch", help="Create a sketch from a set of reads")
    sk.add_argument("-k", "--kmerSize", type=int, default=21)
    sk.add_argument("-o", "--outFile", default="./hulk-" + time.strftime("%Y%m%d%H%M%S"))
    sk.add_argument("-p", "--processors", type=int, default=1)
    sk.add_argument("-f", "--fastq", action="append", default=[],
                    type=lambda s: s.split(","), help="FASTQ file(s) to sketch (comma separated or repeated)")
    sk.add_argument("--fasta", action="store_true")
    sk.add_argument("-w", "--windowSize", type=int, default=9)
    sk.add_argument("-i", "--interval", type=int, default=0)
    sk.add_argument("-s", "--sketchSize", type=int, default=50)
    sk.add_argument("-x", "--decayRatio", type=float, default=1.0)
    sk.add_argument("-b", "--bannerLabel", default="blank")
    sk.add_argument("--device", type=int, default=0)
    sm = sub.add_parser("smash", help="Smash a bunch of sketches and return a similarity matrix")
    sm.add_argument("-k", "--kmerSize", type=int, default=21)
    sm.add_argument("-o", "--outFile", default="./hulk-" + time.strftime("%Y%m%d%H%M%S"))
    sm.add_argument("-p", "--processors", type=int, default=1)
    sm.add_argument("-d", "--sketchDir", default="./")
    sm.add_argument("--recursive", action="store_true")
    sm.add_argument("-a", "--algorithm", default="histosketch")
    sm.add_argument("-m", "--metric", default="jaccard")
    sm.add_argument("--bannerMatrix", action="store_true")
    sm.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    try:
        if a.cmd == "smash":
            from .smash import smash
            log(f"this is hulk (version {VERSION})")
            log("starting the smash subcommand")
            log("checking parameters and collecting sketches...")
            log(f"\talgorithm: {a.algorithm}")
            log(f"\tk-mer size: {a.kmerSize}")
            log(f"\tcreate matrix for banner: {'true' if a.bannerMatrix else 'false'}")
            order, _ = smash(a.sketchDir, a.outFile, a.kmerSize, a.algorithm, a.metric, a.recursive, a.device,
                             banner_matrix=a.bannerMatrix)
            log(f"\tnumber of sketch objects: {len(order)}")
            log("HULK SMASH!")
            log(f"\twritten similarity matrix to disk: {a.outFile}.hulk-matrix.csv")
            if a.bannerMatrix:
                log(f"\twritten banner matrix to disk: {a.outFile}.banner-matrix.csv")
            log("finished")
            return 0
        a.fastq = [f for grp in a.fastq for f in grp if f]
        run_sketch(a)
    except HulkError as e:
        log(f"ERROR---> {e.message}")        # helpers.ErrorCheck -> log.Fatalf (helpers.go:31-35)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
