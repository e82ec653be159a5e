// The cgo binding a maintainer of will-rowe/hulk would add to call libhulkhip (include/hulk_hip.h) from src/pipeline:
// reviewable source, NOT built or run in this repository (no Go toolchain in the image).  It is the first Go block of
// INTEGRATION.md §2, word for word (tests/test_abi_and_host.py keeps the two identical).

// Package gpusketch binds libhulkhip.so (include/hulk_hip.h).
package gpusketch

/*
#cgo CFLAGS: -I${SRCDIR}/../../third_party/hulk_hip/include
#cgo LDFLAGS: -L${SRCDIR}/../../third_party/hulk_hip/lib -lhulkhip
#include <stdlib.h>
#include "hulk_hip.h"
*/
import "C"

import (
	"errors"
	"unsafe"
)

// abiVersion is the HULK_ABI_VERSION this file was written against; New refuses another library.
const abiVersion = 4

// Sketcher plays the role of theBoss + the Sketcher's HistoSketch for one run.
type Sketcher struct {
	ctx     *C.hulk_ctx
	k, s    uint
	bins    int32
	bases   []byte   // batch staging: sequences are copied here (no Go pointer is retained by C)
	offsets []uint64
	batch   int
	// multi-GPU (Shard): this process is rank `rank` of `world`, one GPU each
	sharded     bool
	rank, world uint
	interval    uint
}

// New = findMinimizers (boss.go:54) + histosketch.NewHistoSketch (histosketch.go:50).
func New(k, w, sketchSize uint, bins int32, decay float64, interval uint, device int) (*Sketcher, error) {
	return NewRank(k, w, sketchSize, bins, decay, interval, device, 0, 1)
}

// NewRank is New for rank `rank` of a `world`-GPU run: the context owns the sketch slots
// [S*rank/world, S*(rank+1)/world) (count-min is replicated, the CWS update is slot-sharded).
func NewRank(k, w, sketchSize uint, bins int32, decay float64, interval uint, device int, rank, world uint) (*Sketcher, error) {
	if v := int(C.hulk_abi_version()); v != abiVersion {
		return nil, errors.New("libhulkhip.so has another ABI version than this binding")
	}
	p := C.hulk_params{k: C.uint32_t(k), w: C.uint32_t(w), sketch_size: C.uint32_t(sketchSize),
		num_bins: C.int32_t(bins), decay_ratio: C.double(decay), interval: C.uint32_t(interval),
		device: C.int32_t(device)}
	if world > 1 {
		lo, hi := sketchSize*rank/world, sketchSize*(rank+1)/world
		p.slot_begin, p.slot_count = C.uint32_t(lo), C.uint32_t(hi-lo)
	}
	var ctx *C.hulk_ctx
	if rc := C.hulk_create(&p, &ctx); rc != C.HULK_OK {
		return nil, errors.New(C.GoString(C.hulk_last_error(nil))) // same text the reference logs
	}
	return &Sketcher{ctx: ctx, k: k, s: sketchSize, bins: bins, offsets: []uint64{0}, batch: 1 << 16,
		rank: rank, world: world, interval: interval}, nil
}

// UniqueID draws the 128-byte RCCL id on ONE rank; the host hands it to the others (any channel).
func UniqueID() ([]byte, error) {
	id := make([]byte, C.HULK_UNIQUE_ID_BYTES)
	if rc := C.hulk_comm_unique_id(unsafe.Pointer(&id[0])); rc != C.HULK_OK {
		return nil, errors.New(C.GoString(C.hulk_last_error(nil)))
	}
	return id, nil
}

// Shard connects the ranks (RCCL over xGMI, inside the library; collective call).  Afterwards AddSeq takes THIS RANK's
// reads: of every step of world*T sketching intervals of the global stream (T = hulk_batch_size) the whole intervals
// [rank*T, (rank+1)*T), in stream order; a full share (T*interval reads) crosses into C as one hulk_step_sharded_host.
// The interval rule stays sketch.go:211-215 on the GLOBAL stream: the sketch is the single-GPU one.
func (g *Sketcher) Shard(id []byte) error {
	if g.interval == 0 || len(id) != C.HULK_UNIQUE_ID_BYTES {
		return errors.New("a sharded run needs interval > 0 and a HULK_UNIQUE_ID_BYTES id")
	}
	if err := g.err(C.hulk_comm_init(g.ctx, unsafe.Pointer(&id[0]), C.uint32_t(g.rank), C.uint32_t(g.world))); err != nil {
		return err
	}
	g.sharded = true
	g.batch = int(C.hulk_batch_size(g.ctx)) * int(g.interval)
	return nil
}

func (g *Sketcher) pushStep(stepIntervals uint) error {
	n := len(g.offsets) - 1
	var b *C.uint8_t
	if n > 0 {
		b = (*C.uint8_t)(unsafe.Pointer(&g.bases[0]))
	}
	rc := C.hulk_step_sharded_host(g.ctx, b, (*C.uint64_t)(unsafe.Pointer(&g.offsets[0])), C.uint64_t(n), C.uint32_t(stepIntervals))
	g.bases, g.offsets = g.bases[:0], g.offsets[:1]
	return g.err(rc)
}

// StopWorkSharded ends a sharded stream: lastStepIntervals = sketching intervals of the GLOBAL stream in the last, ragged
// step (the same value on every rank; 0 if the stream ended on a step border), this rank's remaining reads are its share.
func (g *Sketcher) StopWorkSharded(lastStepIntervals uint) error {
	if lastStepIntervals > 0 {
		if err := g.pushStep(lastStepIntervals); err != nil {
			return err
		}
	}
	return g.err(C.hulk_finish(g.ctx))
}

// AddSeq = theBoss.AddSeq (boss.go:24-26); sequences are batched before crossing into C.
func (g *Sketcher) AddSeq(seq []byte) error {
	g.bases = append(g.bases, seq...)
	g.offsets = append(g.offsets, uint64(len(g.bases)))
	if g.sharded {
		if len(g.offsets)-1 == g.batch {
			return g.pushStep(g.world * uint(C.hulk_batch_size(g.ctx)))
		}
		return nil
	}
	if len(g.offsets) > g.batch {
		return g.push()
	}
	return nil
}

func (g *Sketcher) push() error {
	n := len(g.offsets) - 1
	if n == 0 {
		return nil
	}
	rc := C.hulk_add_reads(g.ctx, (*C.uint8_t)(unsafe.Pointer(&g.bases[0])),
		(*C.uint64_t)(unsafe.Pointer(&g.offsets[0])), C.uint64_t(n))
	g.bases, g.offsets = g.bases[:0], g.offsets[:1]
	return g.err(rc)
}

// Flush = theBoss.Flush (boss.go:34-36).  With params.interval set, the library applies the
// interval rule of sketch.go:211-215 itself and this is only needed for the final flush.
func (g *Sketcher) Flush() error {
	if err := g.push(); err != nil {
		return err
	}
	return g.err(C.hulk_flush(g.ctx))
}

// StopWork = final Flush + theBoss.StopWork (sketch.go:219-224).
func (g *Sketcher) StopWork() error {
	if err := g.push(); err != nil {
		return err
	}
	return g.err(C.hulk_finish(g.ctx))
}

// GetMinimizerCount = theBoss.GetMinimizerCount (boss.go:39-41).
func (g *Sketcher) GetMinimizerCount() int {
	var reads, mins, length C.uint64_t
	C.hulk_get_counters(g.ctx, &reads, &mins, &length)
	return int(mins)
}

// Sketch fills the exported fields sketchio needs (histosketch.go:40-41).
func (g *Sketcher) Sketch() (mins []uint, weights []float64, err error) {
	m := make([]uint64, g.s)
	weights = make([]float64, g.s)
	var rc C.int
	if g.sharded { // all-gather of the ranks' slot shards: the whole sketch on every rank
		rc = C.hulk_gather_sketch(g.ctx, (*C.uint64_t)(unsafe.Pointer(&m[0])), (*C.double)(unsafe.Pointer(&weights[0])))
	} else {
		rc = C.hulk_get_sketch(g.ctx, (*C.uint64_t)(unsafe.Pointer(&m[0])), (*C.double)(unsafe.Pointer(&weights[0])))
	}
	mins = make([]uint, g.s)
	for i, v := range m {
		mins[i] = uint(v)
	}
	return mins, weights, g.err(rc)
}

func (g *Sketcher) Close() { C.hulk_destroy(g.ctx); g.ctx = nil }

func (g *Sketcher) err(rc C.int) error {
	if rc == C.HULK_OK {
		return nil
	}
	return errors.New(C.GoString(C.hulk_last_error(g.ctx)))
}

// SmashFiles = runSmash + makeMatrix (cmd/smash.go:160-226) in one call: LoadHULKdata for every file (JSON, class / version, the MD5 of
// the little-endian mins), FindSketch(kSize, algo), the pairwise matrix on the GPU and <outFile>.hulk-matrix.csv as encoding/csv
// writes it, all inside libhulkhip.  The error text is the one the reference would have logged.
func SmashFiles(jsonFiles []string, kSize uint, algo, metric, matrixCSV string, device int) error {
	cPaths := make([]*C.char, len(jsonFiles))
	for i, f := range jsonFiles {
		cPaths[i] = C.CString(f)
		defer C.free(unsafe.Pointer(cPaths[i]))
	}
	cAlgo, cMetric, cOut := C.CString(algo), C.CString(metric), C.CString(matrixCSV)
	defer C.free(unsafe.Pointer(cAlgo))
	defer C.free(unsafe.Pointer(cMetric))
	defer C.free(unsafe.Pointer(cOut))
	errbuf := make([]byte, 4096)
	var first **C.char
	if len(cPaths) > 0 {
		first = &cPaths[0]
	}
	rc := C.hulk_smash_files(C.int(device), first, C.uint32_t(len(cPaths)), C.uint32_t(kSize), cAlgo, cMetric, 0, cOut, nil, nil, nil,
		(*C.char)(unsafe.Pointer(&errbuf[0])), C.uint64_t(len(errbuf)))
	if rc != C.HULK_OK {
		return errors.New(C.GoString((*C.char)(unsafe.Pointer(&errbuf[0]))))
	}
	return nil
}
