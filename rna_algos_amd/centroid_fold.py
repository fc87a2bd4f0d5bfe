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


def centroid_fold_multi(ctx, basepair_probs, seq_len, centroid_thresholds):
    """The folds of one bpp matrix for several thresholds at once, the Theta(n^3) fill on the GPU
    of `ctx` (rnamc_centroid_fold_multi; what src/bin/centroid_fold.rs:147-161 loops over).
    Bit-identical to centroid_fold per threshold.  Returns a list of CentroidFold."""
    packed = np.ascontiguousarray(basepair_probs.packed, dtype=np.float32)
    g = np.ascontiguousarray(centroid_thresholds, dtype=np.float32)
    maxp = max(seq_len // 2, 1)
    pairs = np.zeros((len(g), maxp, 2), dtype=np.uint32)
    npairs = np.zeros(len(g), dtype=np.uint32)
    acc = np.zeros(len(g), dtype=np.float32)
    _lib.check(_lib.lib().rnamc_centroid_fold_multi(ctx._h, packed.ctypes.data, seq_len, g.ctypes.data,
                                                    len(g), pairs.ctypes.data, maxp,
                                                    npairs.ctypes.data, acc.ctypes.data))
    return [CentroidFold([(int(a), int(b)) for a, b in pairs[x, :npairs[x]]], float(acc[x]))
            for x in range(len(g))]


def get_fold_str(fold, seq_len):
    """src/bin/centroid_fold.rs:197-207"""
    s = [UNPAIR] * seq_len
    for i, j in fold.basepair_pos_pairs:
        s[i] = BASEPAIR_LEFT
        s[j] = BASEPAIR_RIGHT
    return "".join(s)
