"""Counter-based synthetic reads: base (read_idx, pos) depends only on
(seed, read_idx * read_len + pos), so any shard can be generated independently, on the host
(numpy) or in HBM (torch), with identical bytes."""
import numpy as np

SEED = 0x48554C4B  # "HULK"
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_M1, _M2, _G = 0xBF58476D1CE4E5B9, 0x94D049BB133111EB, 0x9E3779B97F4A7C15


def _mix_np(g, seed):
    with np.errstate(over="ignore"):
        z = g + np.uint64((seed * _G) & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_M1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_M2)
        z = z ^ (z >> np.uint64(31))
    return z


def reads_numpy(first_read, n_reads, read_len=150, seed=SEED):
    """(bases uint8[n_reads*read_len], offsets uint64[n_reads+1]) for reads [first_read, +n_reads)."""
    g = np.arange(first_read * read_len, (first_read + n_reads) * read_len, dtype=np.uint64)
    bases = _ACGT[(_mix_np(g, seed) & np.uint64(3)).astype(np.intp)]
    offsets = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len)
    return bases, offsets


def _signed(v):
    v &= 0xFFFFFFFFFFFFFFFF
    return v - (1 << 64) if v >> 63 else v


def reads_torch(first_read, n_reads, read_len=150, seed=SEED, device="cuda", chunk=1 << 24):
    """Same bytes as reads_numpy, generated directly in device memory.
    Returns (bases uint8 tensor, padded to a multiple of 8 bytes + 8; offsets int64 tensor [n+1])."""
    import torch
    total = n_reads * read_len
    bases = torch.empty(((total + 15) // 8) * 8, dtype=torch.uint8, device=device)
    bases[total:] = 0
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    add = _signed(seed * _G)
    start = first_read * read_len

    def lsr(z, s):  # logical shift right on int64
        return (z >> s) & ((1 << (64 - s)) - 1)

    for lo in range(0, total, chunk):
        hi = min(total, lo + chunk)
        z = torch.arange(start + lo, start + hi, dtype=torch.int64, device=device) + add
        z = (z ^ lsr(z, 30)) * _signed(_M1)
        z = (z ^ lsr(z, 27)) * _signed(_M2)
        z = z ^ lsr(z, 31)
        bases[lo:hi] = lut[(z & 3)]
    offsets = torch.arange(n_reads + 1, dtype=torch.int64, device=device) * read_len
    return bases, offsets
