"""Letter-code helpers.

The reference turns FASTA records into a float one-hot array with a per-base
Python loop (sequences.py:20-31, :101-117) and needs Biopython.  The HIP path
only needs the letters: these helpers go from plain strings to one byte per
base (0..3 = A,C,G,T, the map of sequences.py:9-17) and to the reference's
one-hot layout, in vectorised NumPy.  The FASTA reader below is a small
stand-alone parser (the reference delegates to Bio.SeqIO, sequences.py:33-52).
"""
import os

import numpy as np

_LUT = np.full(256, 255, dtype=np.uint8)
for _i, _ch in enumerate("ACGT"):
    _LUT[ord(_ch)] = _i
    _LUT[ord(_ch.lower())] = _i


PROTEIN = "ACDEFGHIKLMNPQRSTVWY"      # the 20 amino acids, alphabetical one-letter codes (an `input_dims=20` model)


def _lut(alphabet):
    if alphabet == "ACGT":
        return _LUT
    if len(set(alphabet.upper())) != len(alphabet) or not 1 <= len(alphabet) <= 64:
        raise Exception("alphabet: 1 to 64 distinct letters")
    lut = np.full(256, 255, dtype=np.uint8)
    for i, ch in enumerate(alphabet):
        lut[ord(ch.upper())] = i
        lut[ord(ch.lower())] = i
    return lut


def seqsToCodes(seqs, alphabet="ACGT"):
    """list of equal-length strings -> uint8 array (n, L) of codes 0..len(alphabet)-1 (DNA by default: 0..3 = A,C,G,T).
    Sequences containing other letters (e.g. N) raise, like the reference
    skips them at read time (sequences.py:47-51).  Another `alphabet` (e.g. sequences.PROTEIN) gives the codes of a
    model built with input_dims=len(alphabet)."""
    seqs = [s if isinstance(s, str) else str(s) for s in seqs]
    if len({len(s) for s in seqs}) > 1:
        raise Exception("all sequences must have the same length")
    raw = np.frombuffer("".join(seqs).encode("ascii"), dtype=np.uint8).reshape(len(seqs), -1)
    codes = _lut(alphabet)[raw]
    if (codes >= len(alphabet)).any():
        raise Exception("sequences may only contain %s" % ", ".join(alphabet))
    return codes


def codesToOneHot(codes, input_dims=4):
    """(n, L) codes -> (n,1,input_dims,L) float32 one-hot, the layout of seqToOneHot (sequences.py:101-117)."""
    codes = np.asarray(codes)
    n, L = codes.shape
    out = np.zeros((n, 1, input_dims, L), dtype=np.float32)
    out[np.arange(n)[:, None], 0, codes, np.arange(L)[None, :]] = 1
    return out


def seqToOneHot(seqs):
    """Drop-in for secomo.seqToOneHot on plain strings (or objects whose str() is the sequence)."""
    return codesToOneHot(seqsToCodes([str(getattr(s, "seq", s)) for s in seqs]))


class FastaRecord(object):
    """Minimal stand-in for the Bio.SeqRecord objects the reference hands around
    (only .id, .description and .seq are used by secomo)."""

    __slots__ = ("id", "description", "seq")

    def __init__(self, id, description, seq):
        self.id = id
        self.description = description
        self.seq = seq

    def __len__(self):
        return len(self.seq)

    def __str__(self):
        return self.seq

    def __repr__(self):
        return "FastaRecord(id=%r, len=%d)" % (self.id, len(self.seq))


def _iterFasta(handle):
    header, chunks = None, []
    for line in handle:
        line = line.strip()
        if not line:
            continue
        if line.startswith(">"):
            if header is not None:
                yield header, "".join(chunks)
            header, chunks = line[1:], []
        elif header is not None:
            chunks.append(line)
    if header is not None:
        yield header, "".join(chunks)


def readSeqsFromFasta(filename):
    """Read a multi-FASTA file; records containing N/n are skipped with the
    reference's message (sequences.py:33-52).  Returns a list of FastaRecord."""
    seqs = []
    with open(filename) as f:
        for header, seq in _iterFasta(f):
            if "N" in seq or "n" in seq:
                print("skip sequence containing N")
                continue
            ident = header.split(None, 1)[0] if header.strip() else ""
            seqs.append(FastaRecord(ident, header, seq))
    return seqs


def writeFasta(records, filename, width=60):
    """Write records as multi-FASTA, 60 columns per line like Bio.SeqIO.write."""
    with open(filename, "w") as f:
        for r in records:
            f.write(">%s\n" % (r.description or r.id))
            s = str(r.seq)
            for i in range(0, len(s), width):
                f.write(s[i:i + width] + "\n")
    return len(records)


def splitTrainingTest(filename, train_test_ratio, num_top_regions=None, randomize=True):
    """Split a FASTA file into <prefix>_train.fa / <prefix>_test.fa
    (sequences.py:54-98: the first int(n*ratio) indices of the permutation are the test set)."""
    seqs = readSeqsFromFasta(filename)
    if num_top_regions:
        seqs = seqs[:num_top_regions]
    if randomize:
        idx_permut = list(np.random.permutation(len(seqs)))
    else:
        idx_permut = list(range(len(seqs)))
    ntest = int(len(seqs) * train_test_ratio)
    itest, itrain = idx_permut[:ntest], idx_permut[ntest:]
    prefix = ".".join(filename.split(".")[:-1])
    writeFasta([seqs[i] for i in itrain], prefix + "_train.fa")
    writeFasta([seqs[i] for i in itest], prefix + "_test.fa")


def fastaToCodes(filename):
    """FASTA file -> uint8 codes (n, L) without building per-base Python objects:
    the packed-ingestion path of SURVEY 8(f)-2 (feed the result to CRBM.fit /
    freeEnergy / motifHitProbs, which upload 1 byte per base)."""
    return seqsToCodes([r.seq for r in readSeqsFromFasta(filename)])


def load_sample(filename=None):
    """One-hot sample data (sequences.py:120-134).  The reference ships an Oct4
    ChIP-seq FASTA inside its package; this package carries no data files, so
    the path must be given (or set CRBM_SAMPLE_FASTA)."""
    filename = filename or os.environ.get("CRBM_SAMPLE_FASTA")
    if not filename or not os.path.exists(filename):
        raise Exception("load_sample needs the path of a FASTA file (argument or CRBM_SAMPLE_FASTA)")
    return seqToOneHot(readSeqsFromFasta(filename))
