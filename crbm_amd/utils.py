"""Motif export.

secomo/utils.py:16-47 writes one file per motif through Bio.motifs.jaspar;
this writer produces the same text layouts without Biopython.  The plotting
helpers of the reference's utils.py (logos, t-SNE, violin plots) are analysis
code outside the hot path and are not rebuilt.
"""
import os

_ALPHABET = ("A", "C", "G", "T")


def formatMotif(pfm, name, fformat="jaspar"):
    """Text of one motif.  'jaspar': header line '>id name' and one bracketed row
    per letter; 'pfm': four bare rows; 'tab': letter<TAB>values rows."""
    rows = [["{0:6.2f}".format(float(v)) for v in pfm[a]] for a in range(4)]
    if fformat == "jaspar":
        lines = [">{0} {1}".format(name, name)]
        lines += ["{0} [{1}]".format(_ALPHABET[a], " ".join(rows[a])) for a in range(4)]
    elif fformat == "pfm":
        lines = [" ".join(rows[a]) for a in range(4)]
    elif fformat == "tab":
        lines = ["\t".join([_ALPHABET[a]] + [t.strip() for t in rows[a]]) for a in range(4)]
    else:
        raise ValueError("Unknown JASPAR format %s" % fformat)
    return "\n".join(lines) + "\n"


def saveMotifs(model, path, name="mot", fformat="jaspar"):
    """Save the model's PFMs, one file <name><i>.pfm per motif (utils.py:16-47;
    motif ids are 1-based in the header, 0-based in the file name, as there)."""
    pfms = model.getPFMs()
    if not os.path.exists(path):
        os.makedirs(path)
    for i, pfm in enumerate(pfms):
        with open(os.path.join(path, "{}{:d}.{}".format(name, i, "pfm")), "w") as f:
            f.write(formatMotif(pfm, name + str(i + 1), fformat))
