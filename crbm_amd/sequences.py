"""Letter-code helpers.

The reference turns FASTA records into a float one-hot array with a per-base
Python loop (sequences.py:20-31, :101-117) and needs Biopython.  The HIP path
only needs the letters: these helpers go from plain strings to one byte per
base (0..3 = A,C,G,T, the map of sequences.py:9-17) and to the reference's
one-hot layout, in vectorised NumPy.  FASTA parsing itself stays out of scope.
"""
import numpy as np

_LUT = np.full(256, 255, dtype=np.uint8)
for _i, _ch in enumerate("ACGT"):
    _LUT[ord(_ch)] = _i
    _LUT[ord(_ch.lower())] = _i


def seqsToCodes(seqs):
    """list of equal-length DNA strings -> uint8 array (n, L) of codes 0..3.
    Sequences containing other letters (e.g. N) raise, like the reference
    skips them at read time (sequences.py:47-51)."""
    seqs = [s if isinstance(s, str) else str(s) for s in seqs]
    if len({len(s) for s in seqs}) > 1:
        raise Exception("all sequences must have the same length")
    raw = np.frombuffer("".join(seqs).encode("ascii"), dtype=np.uint8).reshape(len(seqs), -1)
    codes = _LUT[raw]
    if (codes > 3).any():
        raise Exception("sequences may only contain A, C, G, T")
    return codes


def codesToOneHot(codes):
    """(n, L) codes -> (n,1,4,L) float32 one-hot, the layout of seqToOneHot (sequences.py:101-117)."""
    codes = np.asarray(codes)
    n, L = codes.shape
    out = np.zeros((n, 1, 4, L), dtype=np.float32)
    out[np.arange(n)[:, None], 0, codes, np.arange(L)[None, :]] = 1
    return out


def seqToOneHot(seqs):
    """Drop-in for secomo.seqToOneHot on plain strings (or objects whose str() is the sequence)."""
    return codesToOneHot(seqsToCodes([str(getattr(s, "seq", s)) for s in seqs]))
