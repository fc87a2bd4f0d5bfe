"""ctypes access to the CPU oracle (oracle/librnamc_oracle.so) — test infrastructure.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use this."""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_ORACLE_DIR, "librnamc_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        vp = C.c_void_p
        L.rnamc_oracle_bpp.argtypes = [vp, vp, C.c_uint32, C.c_int, C.c_int, vp, vp]
        L.rnamc_oracle_bpp_dump.argtypes = [vp, vp, C.c_uint32, C.c_int, C.c_int, vp, vp, vp]
        L.rnamc_oracle_fold_sums.argtypes = [vp, vp, C.c_uint32, C.c_int, C.c_int, vp]
        L.rnamc_oracle_bpp_batch.argtypes = [vp, C.c_uint32, vp, vp, C.c_int, C.c_int, vp, vp, vp,
                                             C.c_uint32]
        L.rnamc_oracle_exact_bpp.argtypes = [vp, vp, C.c_uint32, C.c_int, C.c_int, vp, vp]
        L.rnamc_oracle_bruteforce.argtypes = [vp, vp, C.c_uint32, C.c_int, C.c_int, vp, vp, vp]
        L.rnamc_oracle_fold_scores.argtypes = [vp, vp, C.c_uint32, C.c_int, C.c_int, vp, vp, vp,
                                               C.POINTER(vp), C.POINTER(C.c_uint64)]
        L.rnamc_oracle_free.argtypes = [vp]
        L.rnamc_oracle_free.restype = None
        L.rnamc_oracle_centroid_fold.argtypes = [vp, C.c_uint32, C.c_float, vp, C.c_uint32, vp, vp]
        L.rnamc_oracle_durbin.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, vp]
        _lib = L
    return _lib


def _chk(st):
    if st != 0:
        raise RuntimeError(f"oracle status {st}")


def bpp(params_ptr, seq, contra, short=False):
    """-> (packed bpp f32[n(n+1)/2], log partition f32)"""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    n = seq.shape[0]
    out = np.empty(n * (n + 1) // 2, dtype=np.float32)
    logz = np.zeros(1, dtype=np.float32)
    _chk(lib().rnamc_oracle_bpp(params_ptr, seq.ctypes.data, n, int(contra), int(short),
                                out.ctypes.data, logz.ctypes.data))
    return out, np.float32(logz[0])


def exact_bpp(params_ptr, seq, contra, short=False):
    """The same recurrences in f64 with an exact logsumexp (oracle/mccaskill_exact.c; NOT the
    reference's arithmetic) -> (packed bpp f64[n(n+1)/2], ln Z f64)"""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    n = seq.shape[0]
    out = np.empty(n * (n + 1) // 2, dtype=np.float64)
    logz = np.zeros(1, dtype=np.float64)
    _chk(lib().rnamc_oracle_exact_bpp(params_ptr, seq.ctypes.data, n, int(contra), int(short),
                                      out.ctypes.data, logz.ctypes.data))
    return out, float(logz[0])


def bpp_dump(params_ptr, seq, contra, short=False):
    """-> (packed bpp, logz, list of 7 n*n matrices in rnamc_debug_fetch order)"""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    n = seq.shape[0]
    out = np.empty(n * (n + 1) // 2, dtype=np.float32)
    logz = np.zeros(1, dtype=np.float32)
    mats = [np.empty((n, n), dtype=np.float32) for _ in range(7)]
    ptrs = (C.c_void_p * 7)(*[m.ctypes.data for m in mats])
    _chk(lib().rnamc_oracle_bpp_dump(params_ptr, seq.ctypes.data, n, int(contra), int(short),
                                     out.ctypes.data, logz.ctypes.data, ptrs))
    return out, np.float32(logz[0]), mats


FOLD_SUMS_FIELDS = ("sums_external", "sums_rightmost_basepairs_external",
                    "sums_rightmost_basepairs_multibranch", "sums_close", "sums_accessible",
                    "sums_multibranch", "sums_1ormore_basepairs")


def fold_sums(params_ptr, seq, contra, short=False):
    """get_fold_sums / get_fold_sums_contra (src/mccaskill_algo.rs:282, 380) -> dict of the seven
    FoldSums members as n*n f32 (sparse maps dense, -inf = absent)"""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    n = seq.shape[0]
    mats = [np.empty((n, n), dtype=np.float32) for _ in range(7)]
    ptrs = (C.c_void_p * 7)(*[m.ctypes.data for m in mats])
    _chk(lib().rnamc_oracle_fold_sums(params_ptr, seq.ctypes.data, n, int(contra), int(short), ptrs))
    return dict(zip(FOLD_SUMS_FIELDS, mats))


def bpp_batch(params_ptr, seqs, contra, short=False, n_threads=1, want_bpp=True):
    lens = np.array([len(s) for s in seqs], dtype=np.uint64)
    offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    bases = np.concatenate([np.asarray(s, dtype=np.uint8) for s in seqs])
    out_offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum(lens * (lens + 1) // 2, out=out_offsets[1:])
    out = np.empty(int(out_offsets[-1]), dtype=np.float32) if want_bpp else None
    logz = np.empty(len(seqs), dtype=np.float32)
    _chk(lib().rnamc_oracle_bpp_batch(params_ptr, len(seqs), bases.ctypes.data,
                                      offsets.ctypes.data, int(contra), int(short),
                                      out.ctypes.data if want_bpp else None,
                                      out_offsets.ctypes.data, logz.ctypes.data, n_threads))
    if not want_bpp:
        return None, logz
    return [out[int(out_offsets[s]):int(out_offsets[s + 1])] for s in range(len(seqs))], logz


TWOLOOP_DTYPE = np.dtype([("i", "<u4"), ("j", "<u4"), ("k", "<u4"), ("l", "<u4"),
                          ("score", "<f4")])


def fold_scores(params_ptr, seq, contra, short=False):
    """-> (hairpin, multibranch_close, accessible packed triangles (NaN = key absent),
    twoloop entries in the reference's insertion order)"""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    n = seq.shape[0]
    hp, mb, ac = (np.empty(n * (n + 1) // 2, dtype=np.float32) for _ in range(3))
    ptr, count = C.c_void_p(), C.c_uint64()
    _chk(lib().rnamc_oracle_fold_scores(params_ptr, seq.ctypes.data, n, int(contra), int(short),
                                        hp.ctypes.data, mb.ctypes.data, ac.ctypes.data,
                                        C.byref(ptr), C.byref(count)))
    if count.value:
        buf = (C.c_char * (count.value * TWOLOOP_DTYPE.itemsize)).from_address(ptr.value)
        tl = np.frombuffer(buf, dtype=TWOLOOP_DTYPE).copy()
    else:
        tl = np.empty(0, dtype=TWOLOOP_DTYPE)
    lib().rnamc_oracle_free(ptr)
    return hp, mb, ac, tl


def bruteforce(params_ptr, seq, contra, short=False):
    """-> (ln Z exact f64, n*n exact pair probabilities f64, number of structures)"""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    n = seq.shape[0]
    logz = np.zeros(1, dtype=np.float64)
    full = np.zeros((n, n), dtype=np.float64)
    cnt = np.zeros(1, dtype=np.uint64)
    _chk(lib().rnamc_oracle_bruteforce(params_ptr, seq.ctypes.data, n, int(contra), int(short),
                                       logz.ctypes.data, full.ctypes.data, cnt.ctypes.data))
    return float(logz[0]), full, int(cnt[0])


def centroid_fold(packed, n, gamma):
    packed = np.ascontiguousarray(packed, dtype=np.float32)
    pairs = np.zeros((max(n // 2, 1), 2), dtype=np.uint32)
    npairs = C.c_uint32()
    acc = C.c_float()
    _chk(lib().rnamc_oracle_centroid_fold(packed.ctypes.data, n, C.c_float(gamma),
                                          pairs.ctypes.data, pairs.shape[0], C.byref(npairs),
                                          C.byref(acc)))
    return [(int(a), int(b)) for a, b in pairs[:npairs.value]], float(acc.value)


def durbin(scores_ptr, a, b):
    """match probabilities of one pair (sequences WITH pseudo bases) -> f32[len(a), len(b)]"""
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    out = np.empty((len(a), len(b)), dtype=np.float32)
    _chk(lib().rnamc_oracle_durbin(scores_ptr, a.ctypes.data, len(a), b.ctypes.data, len(b),
                                   out.ctypes.data))
    return out


def splitmix_seq(n, seed):
    """SplitMix64 stream, base = top 2 bits of each output."""
    mask = (1 << 64) - 1
    state = seed & mask
    out = np.empty(n, dtype=np.uint8)
    for x in range(n):
        state = (state + 0x9E3779B97F4A7C15) & mask
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
        z = z ^ (z >> 31)
        out[x] = z >> 62
    return out
