// dump_cws S k : writes the CWS parameter matrices r, c, b (each S x k^4 float64, row-major, little
// endian, in that order) that histosketch.NewHistoSketch would build — same generators, same seed,
// same draw order as src/histosketch/histosketch.go:95-126 (the fields themselves are unexported).
package main

import (
	"bufio"
	"encoding/binary"
	"fmt"
	"math"
	"os"
	"strconv"

	rng "github.com/leesper/go_rng"
)

func main() {
	if len(os.Args) != 3 {
		fmt.Fprintln(os.Stderr, "usage: dump_cws <sketchSize> <k>")
		os.Exit(2)
	}
	s, _ := strconv.Atoi(os.Args[1])
	k, _ := strconv.Atoi(os.Args[2])
	bins := k * k * k * k
	n := s * bins
	r, c, b := make([]float64, n), make([]float64, n), make([]float64, n)
	gamma := rng.NewGammaGenerator(1)     // DISTRIBUTION_SEED, histosketch.go:21
	uniform := rng.NewUniformGenerator(1) // a second generator with the same seed
	for i := 0; i < n; i++ {              // slot-major, bin-minor
		r[i] = gamma.Gamma(2, 1)
		c[i] = math.Log(gamma.Gamma(2, 1))
		b[i] = uniform.Float64Range(0, 1) * r[i]
	}
	w := bufio.NewWriterSize(os.Stdout, 1<<20)
	defer w.Flush()
	for _, m := range [][]float64{r, c, b} {
		if err := binary.Write(w, binary.LittleEndian, m); err != nil {
			fmt.Fprintln(os.Stderr, err)
			os.Exit(1)
		}
	}
}
