"""Host-side mirror of the reference's `utils` items that the McCaskill path uses
(reference: src/utils.rs): `bytes2seq` (562-577), the `FoldScoreSets` struct
(91-119) with its constructor / accumulate / transfer (src/mccaskill_algo.rs:24-211),
probability bounds (127-129) and the example FASTA path (126).

`FoldScoreSets` here wraps one `rnamc_params` block of the C ABI: the CONTRAfold
set proper (fields named exactly as in the reference) plus the Turner-2004
constants the reference reads straight from the `rna-ss-params` crate.
"""
import ctypes as C
import os

import numpy as np

from . import _lib

EPSILON = 0.001
PROB_BOUND_LOWER = -EPSILON
PROB_BOUND_UPPER = 1.0 + EPSILON
EXAMPLE_FASTA_FILE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                       "tests", "golden", "sampled_trnas.fa")
A, C_, G, U = 0, 1, 2, 3
PSEUDO_BASE = U + 1
MAX_SEQ_LEN = 65535

_SHAPES = {
    "stack_scores": (4, 4, 4, 4), "terminal_mismatch_scores": (4, 4, 4, 4),
    "dangling_scores_left": (4, 4, 4), "dangling_scores_right": (4, 4, 4),
    "helix_close_scores": (4, 4), "basepair_scores": (4, 4),
    "interior_scores_explicit": (4, 4), "interior_scores_1x1": (4, 4),
    "terminal_mismatch_scores_hairpin": (4, 4, 4, 4),
    "terminal_mismatch_scores_1xmany": (4, 4, 4, 4), "terminal_mismatch_scores_2x3": (4, 4, 4, 4),
    "terminal_mismatch_scores_interior": (4, 4, 4, 4),
    "terminal_mismatch_scores_multibranch": (4, 4, 4, 4),
    "dangling_scores_5prime": (4, 4, 4), "dangling_scores_3prime": (4, 4, 4),
}
_TURNER_SHAPES = {"interior_scores_1x1": (4,) * 6, "interior_scores_1x2": (4,) * 7,
                  "interior_scores_2x2": (4,) * 8}


def bytes2seq(x):
    """ASCII ACGUacgu -> base codes 0..3 (np.uint8).  Any other byte raises
    (the reference panics, src/utils.rs:570-572)."""
    if isinstance(x, str):
        x = x.encode("ascii")
    raw = np.frombuffer(bytes(x), dtype=np.uint8)
    out = np.empty(raw.shape[0], dtype=np.uint8)
    _lib.check(_lib.lib().rnamc_bytes2seq(raw.ctypes.data, raw.shape[0], out.ctypes.data))
    return out


def read_fasta(path):
    """Minimal FASTA reader standing in for bio::io::fasta::Reader: list of
    (id, seq codes)."""
    recs, cur_id, cur = [], None, []
    with open(path, "rb") as fh:
        for line in fh:
            line = line.strip()
            if not line:
                continue
            if line.startswith(b">"):
                if cur_id is not None:
                    recs.append((cur_id, bytes2seq(b"".join(cur))))
                cur_id = line[1:].split()[0].decode() if len(line) > 1 else ""
                cur = []
            else:
                cur.append(line)
    if cur_id is not None:
        recs.append((cur_id, bytes2seq(b"".join(cur))))
    return recs


class FoldScoreSets:
    """`FoldScoreSets::new(init_val)` then `.transfer()` as every reference caller
    does (tests/tests.rs:21-22, src/bin/mccaskill_algo.rs:59-60)."""

    def __init__(self, init_val=0.0, _buf=None):
        L = _lib.lib()
        self._buf = _buf if _buf is not None else np.zeros(L.rnamc_params_sizeof(), dtype=np.uint8)
        if _buf is None:
            _lib.check(L.rnamc_params_new(C.c_float(init_val), self._buf.ctypes.data))
        self._fields = {}
        name, off, cnt = C.c_char_p(), C.c_uint64(), C.c_uint64()
        idx = 0
        while L.rnamc_params_field(idx, C.byref(name), C.byref(off), C.byref(cnt)) == _lib.OK:
            self._fields[name.value.decode()] = (off.value, cnt.value)
            idx += 1

    # -- constructors ------------------------------------------------------
    @classmethod
    def new(cls, init_val=0.0):
        return cls(init_val)

    @classmethod
    def synthetic(cls, seed=0):
        """Seeded synthetic tables for both models (the real ones live in the
        absent rna-ss-params crate)."""
        L = _lib.lib()
        buf = np.zeros(L.rnamc_params_sizeof(), dtype=np.uint8)
        _lib.check(L.rnamc_params_synthetic(C.c_uint64(seed), buf.ctypes.data))
        return cls(_buf=buf)

    @classmethod
    def load(cls, path):
        L = _lib.lib()
        buf = np.zeros(L.rnamc_params_sizeof(), dtype=np.uint8)
        _lib.check(L.rnamc_params_load(os.fsencode(path), buf.ctypes.data))
        return cls(_buf=buf)

    def save(self, path):
        _lib.check(_lib.lib().rnamc_params_save(self.ptr, os.fsencode(path)))

    # -- reference methods -------------------------------------------------
    def accumulate(self):
        off, _ = self._fields["contra.hairpin_scores_len"]
        _lib.check(_lib.lib().rnamc_fold_score_sets_accumulate(self._buf.ctypes.data + off))

    def transfer(self, source=None):
        """Copy the compiled tables (of `source`, default: the table set named by
        $RNAMC_TABLES / set_default_tables(); raises NoTablesError when neither is given) the
        way the reference's transfer() copies the crate constants; Turner constants are taken
        over as they are."""
        src = source if source is not None else default_tables()
        off, _ = self._fields["contra.hairpin_scores_len"]
        # Turner block and header come over wholesale; the contra block goes through
        # the canonical-pair mask of transfer().
        mine = self._buf[off:].copy()
        self._buf[:] = src._buf
        self._buf[off:] = mine
        _lib.check(_lib.lib().rnamc_fold_score_sets_transfer(
            self._buf.ctypes.data + off, src._buf.ctypes.data + off))

    # -- plumbing ------------------------------------------------------------
    @property
    def ptr(self):
        return self._buf.ctypes.data

    def content_key(self):
        """Digest of the whole parameter block: host mirrors re-upload the tables of a
        cached device context when this changes (the set is mutable, the reference reads it
        on every call)."""
        import hashlib
        return hashlib.blake2b(self._buf.tobytes(), digest_size=16).digest()

    @property
    def is_synthetic(self):
        """table_id of the header: rnamc_params_synthetic tags it "SYNT" ^ seed."""
        return (int(self._buf[8:16].view(np.uint64)[0]) >> 32) == 0x53594E54

    def field(self, qualified):
        off, cnt = self._fields[qualified]
        arr = self._buf[off:off + 4 * cnt].view(np.float32)
        short = qualified.split(".", 1)[1]
        if qualified.startswith("turner.") and short in _TURNER_SHAPES:
            return arr.reshape(_TURNER_SHAPES[short])
        if short in _SHAPES:
            return arr.reshape(_SHAPES[short])
        return arr

    def __getattr__(self, item):
        fields = self.__dict__.get("_fields", {})
        if "contra." + item in fields:
            v = self.field("contra." + item)
            return v if v.shape[0] > 1 or v.ndim > 1 else v
        raise AttributeError(item)

    def turner(self, item):
        return self.field("turner." + item)


_default = None


class NoTablesError(RuntimeError):
    pass


def set_default_tables(tables):
    """Name the table set `transfer()` draws from (a FoldScoreSets holding the "compiled"
    constants): FoldScoreSets.load(path) of a real dump, or — explicitly — a synthetic set."""
    global _default
    _default = tables


def default_tables():
    """The compiled constants `transfer()` copies.  In the reference they come from the
    rna-ss-params crate, which this tree does not hold: a table file dumped from that crate
    (bindings/rust/dump_tables.rs) must be named by $RNAMC_TABLES, or a set installed with
    set_default_tables().  There is NO silent synthetic fallback: results computed from
    made-up tables must never look like the reference's."""
    global _default
    if _default is None:
        path = os.environ.get("RNAMC_TABLES")
        if not path:
            raise NoTablesError(
                "no scoring tables configured: set $RNAMC_TABLES to a table file dumped from the "
                "rna-ss-params crate (bindings/rust/dump_tables.rs), or choose synthetic tables "
                "explicitly (utils.set_default_tables(FoldScoreSets.synthetic(seed)); the CLIs "
                "take --synthetic-tables SEED)")
        _default = FoldScoreSets.load(path)
    return _default
