"""Host-side mirror of the reference's `mccaskill_algo` module over the C ABI.

Reference interface (src/mccaskill_algo.rs:247-255):

    pub fn mccaskill_algo<T>(seq, uses_contra_model, allows_short_hairpins,
                             fold_score_sets) -> (SparseProbMat<T>, FoldScores<T>)

Same names, argument meaning and error behaviour (an empty sequence or a byte
outside ACGU raises where the reference panics).  All arithmetic runs in the HIP
kernels of librnamc.so; nothing here computes.
"""
import ctypes as C
import threading

import numpy as np

from . import _lib
from .utils import FoldScoreSets, MAX_SEQ_LEN


def bpp_len(n):
    return n * (n + 1) // 2


def bpp_index(n, i, j):
    """Slot of pair (i, j), i <= j, in the packed diagonal-major triangle."""
    d = j - i
    return d * n - d * (d - 1) // 2 + i


class BppMatrix:
    """One sequence's result: packed triangle of f32, absent pairs hold -1.0.
    `sparse()` gives the reference's SparseProbMat<T> as a dict {(i, j): p}."""

    def __init__(self, n, packed):
        self.n = int(n)
        self.packed = packed

    def __getitem__(self, ij):
        i, j = ij
        return float(self.packed[bpp_index(self.n, i, j)])

    def sparse(self):
        out = {}
        n = self.n
        off = 0
        for d in range(n):
            row = self.packed[off:off + n - d]
            for i in np.nonzero(row >= -0.5)[0]:
                out[(int(i), int(i) + d)] = float(row[i])
            off += n - d
        return out

    def dense(self):
        n = self.n
        m = np.full((n, n), -1.0, dtype=np.float32)
        off = 0
        for d in range(n):
            idx = np.arange(n - d)
            m[idx, idx + d] = self.packed[off:off + n - d]
            off += n - d
        return m


# one rnamc_twoloop_score (include/rnamc.h)
TWOLOOP_DTYPE = np.dtype([("i", "<u4"), ("j", "<u4"), ("k", "<u4"), ("l", "<u4"),
                          ("score", "<f4")])


def _sparse_scores(n, packed):
    out = {}
    x = 0
    for d in range(n):
        row = packed[x:x + n - d]
        for i in np.nonzero(~np.isnan(row))[0]:
            out[(int(i), int(i) + d)] = float(row[i])
        x += n - d
    return out


class FoldScores:
    """FoldScores<T> (src/mccaskill_algo.rs:13-19): hairpin_scores, twoloop_scores,
    multibranch_close_scores, accessible_scores keyed like the reference's hash maps.
    No in-crate caller reads them, so they are materialised on first access
    (rnamc_fold_scores: device sweep for the key sets, host scoring)."""

    def __init__(self, materialise=None):
        self._materialise = materialise
        self._maps = None

    def _get(self, x):
        if self._maps is None:
            if self._materialise is None:
                self._maps = ({}, {}, {}, {})
            else:
                n, hp, mb, ac, tl = self._materialise()
                two = {(int(e["i"]), int(e["j"]), int(e["k"]), int(e["l"])): float(e["score"])
                       for e in tl}
                self._maps = (_sparse_scores(n, hp), two, _sparse_scores(n, mb),
                              _sparse_scores(n, ac))
        return self._maps[x]

    hairpin_scores = property(lambda self: self._get(0))
    twoloop_scores = property(lambda self: self._get(1))
    multibranch_close_scores = property(lambda self: self._get(2))
    accessible_scores = property(lambda self: self._get(3))


class FoldSums:
    """Mirror of `FoldSums<T>` (src/mccaskill_algo.rs:3-11), the value of the reference's first
    stage `get_fold_sums` / `get_fold_sums_contra` (282, 380).  The five dense members are n x n
    f32 arrays with the reference's initial values where it writes nothing (sums_external 0,
    the others -inf); `sums_close` and `sums_accessible`, hash maps in the reference, are
    {(i, j): value} dicts of the finite entries (built on first use; the dense form with -inf =
    absent is `dense["sums_close"]`)."""
    FIELDS = ("sums_external", "sums_rightmost_basepairs_external",
              "sums_rightmost_basepairs_multibranch", "sums_close", "sums_accessible",
              "sums_multibranch", "sums_1ormore_basepairs")
    SPARSE = ("sums_close", "sums_accessible")

    def __init__(self, n, dense):
        self.n = n
        self.dense = dense
        self._sparse = {}

    def __getattr__(self, name):
        if name in FoldSums.SPARSE:
            if name not in self._sparse:
                m = self.dense[name]
                ii, jj = np.nonzero(np.isfinite(m))
                self._sparse[name] = {(int(i), int(j)): float(m[i, j]) for i, j in zip(ii, jj)}
            return self._sparse[name]
        if name in FoldSums.FIELDS:
            return self.dense[name]
        raise AttributeError(name)


class Context:
    """Owns one rnamc_ctx (tables + workspace on one GPU)."""

    def __init__(self, fold_score_sets, device=-1, workspace_bytes=0):
        self._h = C.c_void_p()
        self._owned = True
        self._key = fold_score_sets.content_key()
        _lib.check(_lib.lib().rnamc_ctx_create(fold_score_sets.ptr, device, workspace_bytes,
                                               C.byref(self._h)))

    @classmethod
    def of_pool(cls, pool, idx=0):
        """Context `idx` of a Pool as a Context (owned by the pool: close() leaves it alone)."""
        self = cls.__new__(cls)
        self._h = C.c_void_p(_lib.lib().rnamc_pool_ctx(pool._h, idx))
        if not self._h:
            raise IndexError(idx)
        self._owned = False
        self._key = pool._key
        self._pool = pool  # keeps the owner alive
        return self

    def sync_params(self, fold_score_sets):
        """Upload the tables again if their contents differ from the ones on the device."""
        key = fold_score_sets.content_key()
        if key != self._key:
            _lib.check(_lib.lib().rnamc_ctx_set_params(self._h, fold_score_sets.ptr))
            self._key = key

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            if getattr(self, "_owned", True):
                try:
                    _lib.lib().rnamc_ctx_destroy(self._h)
                except Exception:  # interpreter shutdown: the binding module is already torn down
                    pass
            self._h = C.c_void_p()

    __del__ = close

    def set(self, name, value):
        _lib.check(_lib.lib().rnamc_ctx_set(self._h, name.encode(), int(value)))

    def stats(self):
        st = _lib.BatchStats()
        _lib.check(_lib.lib().rnamc_ctx_stats(self._h, C.byref(st), C.sizeof(st), None))
        return {k: getattr(st, k) for k, _ in st._fields_}

    def bpp_batch(self, seqs, uses_contra_model, allows_short_hairpins):
        """seqs: list of np.uint8 code arrays -> (list of BppMatrix, log partition f32[])."""
        for s in seqs:
            if len(s) == 0:
                raise _lib.RnamcError(_lib.ERR_EMPTY_SEQ)
        lens = np.array([len(s) for s in seqs], dtype=np.uint64)
        offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
        np.cumsum(lens, out=offsets[1:])
        bases = np.concatenate([np.asarray(s, dtype=np.uint8) for s in seqs]) if seqs else \
            np.zeros(0, np.uint8)
        out_offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
        np.cumsum(lens * (lens + 1) // 2, out=out_offsets[1:])
        bpp = np.empty(int(out_offsets[-1]), dtype=np.float32)
        logz = np.empty(len(seqs), dtype=np.float32)
        _lib.check(_lib.lib().rnamc_bpp_batch(
            self._h, len(seqs), bases.ctypes.data, offsets.ctypes.data, int(bool(uses_contra_model)),
            int(bool(allows_short_hairpins)), bpp.ctypes.data, out_offsets.ctypes.data,
            logz.ctypes.data))
        mats = [BppMatrix(int(lens[s]), bpp[int(out_offsets[s]):int(out_offsets[s + 1])])
                for s in range(len(seqs))]
        return mats, logz

    def bpp_batch_into(self, bases, offsets, uses_contra_model, allows_short_hairpins, bpp,
                       out_offsets, log_partition=None):
        """rnamc_bpp_batch on caller-owned host buffers (numpy arrays): H2D + kernels + D2H."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        out_offsets = np.ascontiguousarray(out_offsets, dtype=np.uint64)
        assert bpp.dtype == np.float32 and bpp.flags.c_contiguous
        _lib.check(_lib.lib().rnamc_bpp_batch(
            self._h, len(offsets) - 1, bases.ctypes.data, offsets.ctypes.data,
            int(bool(uses_contra_model)), int(bool(allows_short_hairpins)), bpp.ctypes.data,
            out_offsets.ctypes.data,
            log_partition.ctypes.data if log_partition is not None else None))

    def bpp_batch_device(self, n_seqs, d_bases_ptr, offsets, uses_contra_model,
                         allows_short_hairpins, d_bpp_ptr, out_offsets, d_logz_ptr, stream_ptr):
        """Everything already in HBM; enqueues on `stream_ptr` and returns."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        out_offsets = np.ascontiguousarray(out_offsets, dtype=np.uint64)
        _lib.check(_lib.lib().rnamc_bpp_batch_device(
            self._h, n_seqs, d_bases_ptr, offsets.ctypes.data, int(bool(uses_contra_model)),
            int(bool(allows_short_hairpins)), d_bpp_ptr, out_offsets.ctypes.data, d_logz_ptr,
            stream_ptr))

    def fold_scores_packed(self, seq, uses_contra_model, allows_short_hairpins):
        """-> (n, hairpin, multibranch_close, accessible packed triangles with NaN = key
        absent, twoloop entries as a TWOLOOP_DTYPE array)."""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        n = len(seq)
        if n == 0:
            raise _lib.RnamcError(_lib.ERR_EMPTY_SEQ)
        hp, mb, ac = (np.empty(bpp_len(n), dtype=np.float32) for _ in range(3))
        count = C.c_uint64(0)
        args = (self._h, seq.ctypes.data, n, int(bool(uses_contra_model)),
                int(bool(allows_short_hairpins)), hp.ctypes.data, mb.ctypes.data, ac.ctypes.data)
        _lib.check(_lib.lib().rnamc_fold_scores(*args, None, 0, C.byref(count)))
        tl = np.empty(count.value, dtype=TWOLOOP_DTYPE)
        _lib.check(_lib.lib().rnamc_fold_scores(*args, tl.ctypes.data, count.value,
                                                C.byref(count)))
        return n, hp, mb, ac, tl

    def fold_sums(self, seq, uses_contra_model, allows_short_hairpins):
        """FoldSums of one sequence (rnamc_fold_sums: the inside sweep alone, reference order)
        -> FoldSums"""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        n = int(seq.shape[0])
        if n == 0:
            raise _lib.RnamcError(_lib.ERR_EMPTY_SEQ)
        mats = [np.empty((n, n), dtype=np.float32) for _ in FoldSums.FIELDS]
        _lib.check(_lib.lib().rnamc_fold_sums(self._h, seq.ctypes.data, n, int(uses_contra_model),
                                             int(allows_short_hairpins), *[m.ctypes.data for m in mats]))
        return FoldSums(n, dict(zip(FoldSums.FIELDS, mats)))

    def debug_fetch(self, seq_idx, which, n):
        out = np.empty((n, n), dtype=np.float32)
        _lib.check(_lib.lib().rnamc_debug_fetch(self._h, seq_idx, which, out.ctypes.data))
        return out


class Pool:
    """Owns one rnamc_pool: a device context per listed GPU (devices=None: every visible one);
    `bpp_batch` shards a batch over them inside librnamc (rnamc_bpp_batch_multi), one host
    thread per device, results written straight into the host triangles."""

    def __init__(self, fold_score_sets, devices=None, workspace_bytes=0):
        self._h = C.c_void_p()
        self._key = fold_score_sets.content_key()
        if devices is None:
            arr, n = None, 0
        else:
            arr = (C.c_int * len(devices))(*devices)
            n = len(devices)
        _lib.check(_lib.lib().rnamc_pool_create(fold_score_sets.ptr, arr, n, workspace_bytes,
                                                C.byref(self._h)))

    def __len__(self):
        return int(_lib.lib().rnamc_pool_size(self._h))

    def sync_params(self, fold_score_sets):
        key = fold_score_sets.content_key()
        if key != self._key:
            _lib.check(_lib.lib().rnamc_pool_set_params(self._h, fold_score_sets.ptr))
            self._key = key

    def set(self, name, value):
        _lib.check(_lib.lib().rnamc_pool_set(self._h, name.encode(), int(value)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            try:
                _lib.lib().rnamc_pool_destroy(self._h)
            except Exception:
                pass
            self._h = C.c_void_p()

    __del__ = close

    def bpp_batch(self, seqs, uses_contra_model, allows_short_hairpins):
        for s in seqs:
            if len(s) == 0:
                raise _lib.RnamcError(_lib.ERR_EMPTY_SEQ)
        lens = np.array([len(s) for s in seqs], dtype=np.uint64)
        offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
        np.cumsum(lens, out=offsets[1:])
        bases = np.concatenate([np.asarray(s, dtype=np.uint8) for s in seqs]) if seqs else \
            np.zeros(0, np.uint8)
        out_offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
        np.cumsum(lens * (lens + 1) // 2, out=out_offsets[1:])
        bpp = np.empty(int(out_offsets[-1]), dtype=np.float32)
        logz = np.empty(len(seqs), dtype=np.float32)
        _lib.check(_lib.lib().rnamc_bpp_batch_multi(
            self._h, len(seqs), bases.ctypes.data, offsets.ctypes.data, int(bool(uses_contra_model)),
            int(bool(allows_short_hairpins)), bpp.ctypes.data, out_offsets.ctypes.data,
            logz.ctypes.data))
        mats = [BppMatrix(int(lens[s]), bpp[int(out_offsets[s]):int(out_offsets[s + 1])])
                for s in range(len(seqs))]
        return mats, logz


def shard_plan(lengths, n_shards):
    """rnamc_shard_plan: shard index of every sequence (host only, no device needed)."""
    lens = np.asarray(lengths, dtype=np.uint64)
    offsets = np.zeros(len(lens) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    out = np.zeros(len(lens), dtype=np.uint32)
    _lib.check(_lib.lib().rnamc_shard_plan(len(lens), offsets.ctypes.data, int(n_shards),
                                           out.ctypes.data))
    return out


# ONE pool per process for the module-level functions (workspaces and staging buffers are reused
# across calls); the single-sequence functions use ITS first context, so a process never holds two
# contexts on one device.  Which devices: RNAMC_DEVICES="0,2" lists them; under a one-process-per-GPU
# launch (LOCAL_RANK set: torchrun, bench.py) the rank's own device only — a rank that saw every
# device would open streams, tables and a workspace on all of them and shard its batch across its
# peers' GPUs; otherwise every visible device, as the reference's binary takes every core
# (src/bin/mccaskill_algo.rs:44-48).  The reference reads `&FoldScoreSets` on every call, and the
# set is mutable: the tables are re-uploaded whenever their CONTENT differs from what the contexts
# hold — never keyed by object identity, nothing is kept alive per set.
_ctx = None
_pool = None
_ctx_lock = threading.RLock()


def default_devices():
    """None = every visible device (resolved by rnamc_pool_create), else the list to use."""
    import os
    if os.environ.get("RNAMC_DEVICES"):
        return [int(x) for x in os.environ["RNAMC_DEVICES"].split(",") if x.strip() != ""]
    if os.environ.get("LOCAL_RANK") is not None:
        return [int(os.environ["LOCAL_RANK"])]
    return None


def _pool_for(fold_score_sets):
    global _pool
    with _ctx_lock:
        if _pool is None:
            _pool = Pool(fold_score_sets, default_devices())
        else:
            _pool.sync_params(fold_score_sets)
        return _pool


def _context_for(fold_score_sets):
    global _ctx
    with _ctx_lock:
        pool = _pool_for(fold_score_sets)  # (uploads changed tables to every context, this one included)
        if _ctx is None:
            _ctx = Context.of_pool(pool, 0)
        _ctx._key = pool._key
        return _ctx


def mccaskill_algo_packed(seq, uses_contra_model, allows_short_hairpins, fold_score_sets):
    seq = np.asarray(seq, dtype=np.uint8)
    if seq.shape[0] > MAX_SEQ_LEN:
        raise _lib.RnamcError(_lib.ERR_SEQ_TOO_LONG)
    with _ctx_lock:  # tables of the shared context stay the caller's until the call returns
        mats, logz = _context_for(fold_score_sets).bpp_batch([seq], uses_contra_model,
                                                             allows_short_hairpins)
    return mats[0], float(logz[0])


def mccaskill_algo(seq, uses_contra_model, allows_short_hairpins, fold_score_sets):
    """(SparseProbMat, FoldScores) like the reference (src/mccaskill_algo.rs:247-280)."""
    mat, _ = mccaskill_algo_packed(seq, uses_contra_model, allows_short_hairpins, fold_score_sets)
    seq = np.array(seq, dtype=np.uint8)
    # the maps are filled on first access, with the tables as they were at call time
    frozen = FoldScoreSets(_buf=fold_score_sets._buf.copy())

    def materialise():
        with _ctx_lock:
            return _context_for(frozen).fold_scores_packed(seq, uses_contra_model,
                                                           allows_short_hairpins)
    return mat.sparse(), FoldScores(materialise)


def get_fold_sums(seq, fold_score_sets):
    """`get_fold_sums<T>(seq, &mut fold_scores) -> FoldSums<T>` (src/mccaskill_algo.rs:282-378,
    Turner) on the device.  The reference also fills `fold_scores` on the way; here that is
    `mccaskill_algo(...)[1]` / `Context.fold_scores_packed`."""
    return _context_for(fold_score_sets).fold_sums(seq, False, False)


def get_fold_sums_contra(seq, allows_short_hairpins, fold_score_sets):
    """`get_fold_sums_contra<T>(seq, &mut fold_scores, allows_short_hairpins, fold_score_sets)`
    (src/mccaskill_algo.rs:380-516) on the device."""
    return _context_for(fold_score_sets).fold_sums(seq, True, allows_short_hairpins)


def mccaskill_algo_batch(seqs, uses_contra_model, allows_short_hairpins, fold_score_sets):
    """Whole FASTA at once, over the process's devices (`default_devices`: every visible GPU unless
    RNAMC_DEVICES / LOCAL_RANK say otherwise) — what src/bin/mccaskill_algo.rs:58-93 does on all
    cores with one pool task per record."""
    with _ctx_lock:
        return _pool_for(fold_score_sets).bpp_batch(list(seqs), uses_contra_model,
                                                    allows_short_hairpins)
