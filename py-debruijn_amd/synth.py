"""Counter-based synthetic DNA read generator (host twin of the HIP generator).

The reference's ``I_generateInputReads.py`` is not a read simulator (it filters
mass-spec TSVs and shells out to tools that are not in the repo); what this
module keeps from it is the on-disk format it writes:

* ``{froot}/input_reads.fasta`` -- ``>input_read{i}\\n{seq}\\n``
  (I_generateInputReads.py:107-109)
* ``{froot}/setting.json`` with the keys ``score_cut, k_lowerlimit,
  k_upperlimit, threshold, source`` (I_generateInputReads.py:63-64,71-72)

Every base is a pure function of (seed, counter), so the same reads can be
regenerated on the device (``dbg_synth_reads`` in csrc/dbg_hip.hip) without a
FASTA round trip; ``tests/test_synth.py`` pins the two against one checksum.

Model: genome i.i.d. uniform over ACGT, length G; reads of
fixed length L, start uniform in [0, G-L], forward strand only (the reference
has no reverse-complement logic); optional per-base substitution errors.
"""
from __future__ import annotations

import json
import os

import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
GOLDEN = np.uint64(0x9E3779B97F4A7C15)
C_CTR = np.uint64(0xD6E8FEB86659FD93)
STREAM_GENOME, STREAM_START, STREAM_ERROR = 1, 2, 3
ERR_DENOM_BITS = 24
ALPHABET = np.frombuffer(b"ACGT", dtype=np.uint8)


def mix64(x):
    """splitmix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x ^ (x >> np.uint64(30))
        x = x * np.uint64(0xBF58476D1CE4E5B9)
        x = x ^ (x >> np.uint64(27))
        x = x * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x


def stream_key(seed, stream):
    with np.errstate(over="ignore"):
        return mix64(np.uint64(seed) + np.uint64(stream) * GOLDEN)


def draw(key, ctr):
    """The (key, counter) -> uint64 function shared with the device generator."""
    ctr = np.asarray(ctr, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return mix64(key ^ (ctr * C_CTR))


def err_threshold(err_rate):
    """Error probability as an integer threshold on 24 random bits."""
    return int(round(float(err_rate) * (1 << ERR_DENOM_BITS)))


def genome_codes(seed, start, length):
    """2-bit codes (0..3 -> A,C,G,T) of genome[start:start+length]."""
    ctr = np.arange(start, start + length, dtype=np.uint64)
    return (draw(stream_key(seed, STREAM_GENOME), ctr) & np.uint64(3)).astype(np.uint8)


def read_starts(seed, first_read, n_reads, genome_len, read_len):
    ctr = np.arange(first_read, first_read + n_reads, dtype=np.uint64)
    span = np.uint64(genome_len - read_len + 1)
    return draw(stream_key(seed, STREAM_START), ctr) % span


def reads_ascii(seed, genome_len, n_reads, read_len, err_rate=0.0, first_read=0, chunk=1 << 16):
    """Returns an (n_reads, read_len) uint8 array of ASCII bases.

    ``first_read`` lets a rank generate its own shard of one global read set.
    """
    out = np.empty((n_reads, read_len), dtype=np.uint8)
    thr = np.uint64(err_threshold(err_rate))
    kg = stream_key(seed, STREAM_GENOME)
    ke = stream_key(seed, STREAM_ERROR)
    L = np.uint64(read_len)
    col = np.arange(read_len, dtype=np.uint64)[None, :]
    for lo in range(0, n_reads, chunk):
        hi = min(n_reads, lo + chunk)
        st = read_starts(seed, first_read + lo, hi - lo, genome_len, read_len)[:, None]
        codes = draw(kg, st + col) & np.uint64(3)
        if int(thr) > 0:
            ridx = np.arange(first_read + lo, first_read + hi, dtype=np.uint64)[:, None]
            with np.errstate(over="ignore"):
                e = draw(ke, ridx * L + col)
            is_err = (e & np.uint64((1 << ERR_DENOM_BITS) - 1)) < thr
            sub = np.uint64(1) + ((e >> np.uint64(ERR_DENOM_BITS)) % np.uint64(3))
            codes = np.where(is_err, (codes + sub) & np.uint64(3), codes)
        out[lo:hi] = ALPHABET[codes.astype(np.intp)]
    return out


def reads_list(seed, genome_len, n_reads, read_len, err_rate=0.0, first_read=0):
    arr = reads_ascii(seed, genome_len, n_reads, read_len, err_rate, first_read)
    return [row.tobytes().decode("ascii") for row in arr]


def checksum(ascii_bytes):
    """Order-sensitive 64-bit checksum of a byte buffer (same on host and device).

    sum over i of mix64(byte_i + 256 * i)  (mod 2^64).
    """
    b = np.ascontiguousarray(ascii_bytes).reshape(-1).astype(np.uint64)
    total = np.uint64(0)
    step = 1 << 22
    with np.errstate(over="ignore"):
        for lo in range(0, b.size, step):
            idx = np.arange(lo, min(b.size, lo + step), dtype=np.uint64)
            total = total + mix64(b[lo:lo + step] + np.uint64(256) * idx).sum(dtype=np.uint64)
    return int(total)


def write_fasta(path, reads):
    """I_generateInputReads.py:107-109 format."""
    with open(path, "w") as fh:
        for i, r in enumerate(reads):
            fh.write(f">input_read{i}\n{r}\n")


def write_froot(froot, reads, k_lower, k_upper, threshold=2, score_cut=0.0, source=""):
    """Creates ``{froot}/setting.json`` + ``{froot}/input_reads.fasta``."""
    os.makedirs(froot, exist_ok=True)
    setting = {"score_cut": score_cut, "threshold": threshold, "k_lowerlimit": k_lower,
               "k_upperlimit": k_upper, "source": source}
    with open(os.path.join(froot, "setting.json"), "w") as fh:
        json.dump(setting, fh)
    write_fasta(os.path.join(froot, "input_reads.fasta"), reads)
    return setting
