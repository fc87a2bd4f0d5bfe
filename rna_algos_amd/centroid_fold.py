"""Host-side mirror of the reference's `centroid_fold` module
(src/centroid_fold.rs:4-7,25-105) over rnamc_centroid_fold."""
import ctypes as C

import numpy as np

from . import _lib

UNPAIR, BASEPAIR_LEFT, BASEPAIR_RIGHT = ".", "(", ")"
MIN_POW_2, MAX_POW_2 = -7, 10  # src/bin/centroid_fold.rs:9-10


class CentroidFold:
    def __init__(self, basepair_pos_pairs=None, expect_accuracy=0.0):
        self.basepair_pos_pairs = basepair_pos_pairs or []
        self.expect_accuracy = expect_accuracy


def centroid_fold(basepair_probs, seq_len, centroid_threshold):
    """basepair_probs: BppMatrix (packed triangle).  Returns CentroidFold."""
    packed = np.ascontiguousarray(basepair_probs.packed, dtype=np.float32)
    pairs = np.zeros((max(seq_len // 2, 1), 2), dtype=np.uint32)
    npairs = C.c_uint32()
    acc = C.c_float()
    _lib.check(_lib.lib().rnamc_centroid_fold(packed.ctypes.data, seq_len,
                                              C.c_float(centroid_threshold), pairs.ctypes.data,
                                              pairs.shape[0], C.byref(npairs), C.byref(acc)))
    return CentroidFold([(int(a), int(b)) for a, b in pairs[:npairs.value]], float(acc.value))


def get_fold_str(fold, seq_len):
    """src/bin/centroid_fold.rs:197-207"""
    s = [UNPAIR] * seq_len
    for i, j in fold.basepair_pos_pairs:
        s[i] = BASEPAIR_LEFT
        s[j] = BASEPAIR_RIGHT
    return "".join(s)
