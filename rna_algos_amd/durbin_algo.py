"""Host-side mirror of the reference's `durbin_algo` module over the C ABI.

Reference interface (src/durbin_algo.rs):

    pub struct AlignScores { match2match_score, match2insert_score, insert_extend_score,
                             insert_switch_score, init_match_score, init_insert_score,
                             insert_scores, match_scores }            (4-14)
    impl AlignScores { pub fn new(init_val), pub fn transfer(&mut self) }   (25-58)
    pub fn durbin_algo(seq_pair: &SeqPair, align_scores: &AlignScores) -> ProbMat   (73-77)

Sequences carry PSEUDO_BASE at both ends as the reference's callers build them
(src/bin/durbin_algo.rs:49-51).  All arithmetic runs in the HIP kernels of librnamc.so.
"""
import ctypes as C

import numpy as np

from . import _lib
from .utils import PSEUDO_BASE


class _AlignScoresStruct(C.Structure):
    _fields_ = [("match2match_score", C.c_float), ("match2insert_score", C.c_float),
                ("insert_extend_score", C.c_float), ("insert_switch_score", C.c_float),
                ("init_match_score", C.c_float), ("init_insert_score", C.c_float),
                ("insert_scores", C.c_float * 4), ("match_scores", (C.c_float * 4) * 4)]


class AlignScores:
    """`AlignScores::new(init_val)` then `.transfer()` as every reference caller does
    (tests/tests.rs:61-62, src/bin/durbin_algo.rs:54-55)."""

    def __init__(self, init_val=0.0):
        self._s = _AlignScoresStruct()
        _lib.check(_lib.lib().rnamc_align_scores_new(C.c_float(init_val), C.byref(self._s)))

    @classmethod
    def new(cls, init_val=0.0):
        return cls(init_val)

    def transfer(self):
        _lib.check(_lib.lib().rnamc_align_scores_transfer(C.byref(self._s)))

    @property
    def ptr(self):
        return C.addressof(self._s)

    def __getattr__(self, item):
        s = self.__dict__.get("_s")
        if s is not None and item in dict(_AlignScoresStruct._fields_):
            v = getattr(s, item)
            return np.ctypeslib.as_array(v) if hasattr(v, "_length_") else float(v)
        raise AttributeError(item)

    def set(self, item, value):
        """assign a scalar field or fill an array field (the struct's fields are `pub`)"""
        cur = getattr(self._s, item)
        if hasattr(cur, "_length_"):
            np.ctypeslib.as_array(cur)[...] = value
        else:
            setattr(self._s, item, float(value))


def with_pseudo_bases(seq):
    """seq.insert(0, PSEUDO_BASE); seq.push(PSEUDO_BASE) (src/bin/durbin_algo.rs:49-51)"""
    seq = np.asarray(seq, dtype=np.uint8)
    return np.concatenate([[PSEUDO_BASE], seq, [PSEUDO_BASE]]).astype(np.uint8)


# one device context per process for calls without an explicit one (the reference's callers run
# one durbin_algo per pair from a pool: a context per call would spend its time on streams,
# events and workspace allocation)
_ctx = None


def _default_context():
    global _ctx
    if _ctx is None:
        from .mccaskill_algo import Context
        from .utils import FoldScoreSets
        # the pair-HMM reads none of the folding tables: any parameter block will do
        _ctx = Context(FoldScoreSets.new(0.0))
    return _ctx


def durbin_algo_batch(seqs, pairs, align_scores, ctx=None):
    """seqs: sequences WITH pseudo bases; pairs: list of (a, b) indices.  -> list of ProbMat
    (np.float32 arrays of shape (len(a), len(b))), what src/bin/durbin_algo.rs:55-75 computes
    with one pool task per pair."""
    if ctx is None:
        ctx = _default_context()
    seqs = [np.ascontiguousarray(s, dtype=np.uint8) for s in seqs]
    lens = np.array([len(s) for s in seqs], dtype=np.uint64)
    offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    bases = np.concatenate(seqs) if seqs else np.zeros(0, np.uint8)
    pa = np.array([p[0] for p in pairs], dtype=np.uint32)
    pb = np.array([p[1] for p in pairs], dtype=np.uint32)
    sizes = np.array([int(lens[a]) * int(lens[b]) for a, b in pairs], dtype=np.uint64)
    out_offsets = np.zeros(len(pairs) + 1, dtype=np.uint64)
    np.cumsum(sizes, out=out_offsets[1:])
    out = np.empty(int(out_offsets[-1]), dtype=np.float32)
    _lib.check(_lib.lib().rnamc_durbin_batch(
        ctx._h, align_scores.ptr, len(seqs), bases.ctypes.data, offsets.ctypes.data, len(pairs),
        pa.ctypes.data, pb.ctypes.data, out.ctypes.data, out_offsets.ctypes.data))
    return [out[int(out_offsets[p]):int(out_offsets[p + 1])].reshape(int(lens[a]), int(lens[b]))
            for p, (a, b) in enumerate(pairs)]


def durbin_algo(seq_pair, align_scores, ctx=None):
    """ProbMat of one pair (src/durbin_algo.rs:73-77)."""
    return durbin_algo_batch([seq_pair[0], seq_pair[1]], [(0, 1)], align_scores, ctx)[0]
