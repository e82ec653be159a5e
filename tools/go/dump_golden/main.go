// dump_golden reads.fq[.gz] k w sketchSize [interval [decay]] : the reference's own packages, driven
// serially, on a FASTQ file; prints {"histogram_sha256", "n_minimizers", "mins", "weights"} as JSON.
// Line handling is the plain 4-line case (the fixture has no empty lines).
package main

import (
	"bufio"
	"compress/gzip"
	"crypto/sha256"
	"encoding/binary"
	"encoding/hex"
	"encoding/json"
	"fmt"
	"io"
	"os"
	"strconv"
	"strings"

	"github.com/will-rowe/hulk/src/helpers"
	"github.com/will-rowe/hulk/src/histosketch"
	"github.com/will-rowe/hulk/src/kmerspectrum"
	"github.com/will-rowe/hulk/src/minimizer"
)

func die(err error) {
	if err != nil {
		fmt.Fprintln(os.Stderr, "ERROR--->", err)
		os.Exit(1)
	}
}

func main() {
	if len(os.Args) < 5 {
		fmt.Fprintln(os.Stderr, "usage: dump_golden <reads.fq[.gz]> <k> <w> <sketchSize> [interval [decayRatio]]")
		os.Exit(2)
	}
	k, _ := strconv.Atoi(os.Args[2])
	w, _ := strconv.Atoi(os.Args[3])
	s, _ := strconv.Atoi(os.Args[4])
	interval, decay := 0, 1.0
	if len(os.Args) > 5 {
		interval, _ = strconv.Atoi(os.Args[5])
	}
	if len(os.Args) > 6 {
		decay, _ = strconv.ParseFloat(os.Args[6], 64)
	}
	bins := int32(helpers.Pow(uint(k), 4)) // cmd/sketch.go:118
	fh, err := os.Open(os.Args[1])
	die(err)
	var rd io.Reader = fh
	if strings.HasSuffix(os.Args[1], ".gz") {
		gz, err := gzip.NewReader(fh)
		die(err)
		rd = gz
	}
	spectrum, err := kmerspectrum.NewKmerSpectrum(bins)
	die(err)
	hs, err := histosketch.NewHistoSketch(uint(k), uint(s), bins, decay)
	die(err)
	hist := sha256.New()
	flush := func() {
		if spectrum.Cardinality() == 0 {
			return
		}
		dump, err := spectrum.Dump() // "not used yet" below 1 % used bins
		die(err)
		for bin := range dump {
			_ = binary.Write(hist, binary.LittleEndian, uint32(bin.Frequency))
			die(hs.AddElement(uint64(bin.BinID), bin.Frequency))
		}
		spectrum.Wipe()
	}
	nMin, seqCount, line := 0, 0, 0
	sc := bufio.NewScanner(rd)
	for sc.Scan() {
		line++
		if line%4 != 2 {
			continue
		}
		seq := append([]byte(nil), sc.Bytes()...)
		ms, err := minimizer.NewMinimizerSketch(uint(k), uint(w), seq)
		die(err)
		for m := range ms.GetMinimizers() {
			die(spectrum.AddHash(m.(uint64)))
			nMin++
		}
		seqCount++
		if interval != 0 && seqCount%interval == 0 { // pipeline/sketch.go:211
			flush()
		}
	}
	die(sc.Err())
	flush()
	out := map[string]interface{}{
		"n_reads": seqCount, "n_minimizers": nMin, "mins": hs.Sketch, "weights": hs.SketchWeights,
		"histogram_sha256_of_flushed_counts": hex.EncodeToString(hist.Sum(nil)),
	}
	enc := json.NewEncoder(os.Stdout)
	die(enc.Encode(out))
}
