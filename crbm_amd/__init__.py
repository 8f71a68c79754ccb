"""crbm_amd -- MI355X-native training hot path of SECOMO's convolutional RBM.

`CRBM` keeps the reference's Python API (secomo/__init__.py:1,
secomo/convRBM.py:25-726); the arithmetic runs in hand-written HIP kernels
behind the C-ABI in include/crbm_amd.h.
"""
from .crbm import CRBM  # noqa: F401
from . import dist  # noqa: F401
from .sequences import (seqsToCodes, codesToOneHot, seqToOneHot, readSeqsFromFasta,  # noqa: F401
                        splitTrainingTest, fastaToCodes, writeFasta, load_sample)
from .utils import saveMotifs  # noqa: F401

__version__ = "0.1.0"
