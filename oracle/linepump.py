"""CPU restatement of the reference's stream pump — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows, line by line:
  * DataStreamer.Run       src/pipeline/sketch.go:40-79   (bufio.Scanner over STDIN / files, gzip by
                           extension, every line copied with append([]byte(nil), ...) so that an
                           EMPTY line travels as a nil slice)
  * bufio.Scanner/ScanLines (Go standard library, go1.12): '\n' delimited, ONE trailing '\r' dropped,
                           final unterminated line is a token, 64 KiB buffer -> a line of >= 65536
                           bytes ends the scan with "bufio.Scanner: token too long" (log.Fatal)
  * FastqHandler.Run       src/pipeline/sketch.go:99-161  (the four nil-tested line slots; FASTA branch)
  * seqio.NewFASTQread     src/seqio/seqio.go:38-40       (the '@' check, made when the 4th line arrives)

Pinned against: nothing executable (no Go toolchain here) — the reference has no test for this code;
the semantics are read off the source.  The reference's own fixture (tests/golden/test-reads-small.fq.gz,
1000 records) is the only real-data check.
"""
import gzip
import sys

MAX_TOKEN = 64 * 1024


class PumpError(Exception):
    pass


def scan_lines(data: bytes):
    """bufio.Scanner with ScanLines over one input; raises PumpError for an over-long line."""
    pos, n = 0, len(data)
    while pos < n:
        i = data.find(b"\n", pos)
        end = n if i < 0 else i
        if end - pos >= MAX_TOKEN:
            raise PumpError("bufio.Scanner: token too long")
        tok = data[pos:end]
        if tok.endswith(b"\r"):
            tok = tok[:-1]
        yield tok
        pos = end + 1


def data_streamer(paths):
    """sketch.go:40-79 — one item per line; an empty line is None (the nil slice)."""
    if not paths:
        for tok in scan_lines(sys.stdin.buffer.read()):
            yield tok if len(tok) else None
        return
    for p in paths:
        with open(p, "rb") as fh:
            raw = fh.read()
        if p.split(".")[-1] == "gz":
            raw = gzip.decompress(raw)
        for tok in scan_lines(raw):
            yield tok if len(tok) else None


def fastq_handler(lines, fasta):
    """sketch.go:99-161 — yields the Seq handed to theBoss.AddSeq (None for a nil sequence)."""
    l1 = l2 = l3 = l4 = None
    if fasta:
        for line in lines:
            if line is None:                      # len(line) == 0 -> break
                break
            if line[0] == 62:
                if l1 is not None:
                    yield l2
                l1, l2 = line, None
            else:
                l2 = (l2 or b"") + line
        if l1 is None:
            raise PumpError("fasta input holds no header line")   # the reference panics on l1[0] = 64
        yield l2
        return
    for line in lines:
        if l1 is None:
            l1 = line
        elif l2 is None:
            l2 = line
        elif l3 is None:
            l3 = line
        elif l4 is None:
            l4 = line                             # an empty 4th line completes the record too
            if l1[0] != 64:
                raise PumpError("read ID in fastq file does not begin with @: " + l1.decode("latin-1"))
            yield l2
            l1 = l2 = l3 = l4 = None


def sequences(paths, fasta=False):
    return list(fastq_handler(data_streamer(paths), fasta))
