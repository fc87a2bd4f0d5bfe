"""The Rust side of the boundary cannot be compiled in this image (no cargo), so it is
checked mechanically: every `fn` of the shim's `extern "C"` block must be declared in
include/rnamc.h with the same arity, integer widths, float-ness, pointer-ness and
constness; the dumper's field table must reproduce the layout of `rnamc_params`."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "bindings", "rust", "mccaskill_algo.rs")
DUMPER = os.path.join(ROOT, "bindings", "rust", "dump_tables.rs")
HEADER = os.path.join(ROOT, "include", "rnamc.h")

# opaque / struct pointees that correspond across the boundary
OPAQUE = {"rnamc_ctx": "ctx", "RnamcCtx": "ctx", "rnamc_pool": "pool", "RnamcPool": "pool", "rnamc_params": "void", "c_void": "void",
          "void": "void", "rnamc_twoloop_score": "twoloop", "TwoloopScore": "twoloop",
          "rnamc_fold_score_sets": "void", "rnamc_batch_stats": "stats",
          "rnamc_align_scores": "align", "AlignScoresC": "align"}
C_SCALAR = {"int": "i32", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "float": "f32",
            "size_t": "usize", "uint8_t": "u8", "char": "i8", "double": "f64"}
R_SCALAR = {"c_int": "i32", "u32": "u32", "u64": "u64", "i64": "i64", "f32": "f32",
            "usize": "usize", "u8": "u8", "c_char": "i8", "f64": "f64"}


def c_type(t):
    """'const uint8_t*' -> ('ptr', const?, pointee...) / scalar tag"""
    t = t.strip()
    stars = t.count("*")
    t = t.replace("*", " ")
    toks = [x for x in t.split() if x not in ("struct",)]
    const = "const" in toks
    toks = [x for x in toks if x != "const"]
    base = toks[0]
    tag = C_SCALAR.get(base) or OPAQUE.get(base)
    assert tag, f"unknown C type {base!r}"
    out = tag
    for lvl in range(stars):
        # constness of the innermost pointee is what `*const` / `*mut` express
        out = ("ptr", const if lvl == 0 else False, out)
    return out


def rust_type(t):
    t = t.strip()
    m = re.match(r"\*(const|mut)\s+(.*)", t)
    if m:
        inner = rust_type(m.group(2))
        # C `const T**` has a const innermost pointee: Rust spells that on the inner pointer
        return ("ptr", m.group(1) == "const", inner)
    tag = R_SCALAR.get(t) or OPAQUE.get(t)
    assert tag, f"unknown Rust type {t!r}"
    return tag


def normalise(t):
    """Compare pointer trees level by level; for a pointer to pointer C puts `const` on the
    innermost pointee while Rust marks each level: only the innermost level is compared."""
    if isinstance(t, tuple):
        _, const, inner = t
        if isinstance(inner, tuple):
            return ("ptr", normalise(inner))
        return ("ptr", "const" if const else "mut", inner)
    return t


def c_decls():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(rnamc_\w+)\s*\(([^;{}]*?)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        if "typedef" in ret:
            continue
        args = [a.strip() for a in args.split(",")] if args.strip() not in ("", "void") else []
        params = []
        for a in args:
            mm = re.match(r"(.*?)(\w+)$", a)  # split off the parameter name
            params.append(c_type(mm.group(1)))
        decls[name] = (c_type(ret) if ret.strip() != "void" else "unit", params)
    return decls


DURBIN_SHIM = os.path.join(ROOT, "bindings", "rust", "durbin_algo.rs")


def rust_decls(path=SHIM):
    src = open(path).read()
    src = re.sub(r"//[^\n]*", "", src)
    block = re.search(r'extern "C" \{(.*?)\n\}', src, flags=re.S).group(1)
    decls = {}
    for m in re.finditer(r"fn\s+(\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2), m.group(3)
        params = []
        for a in [x for x in (y.strip() for y in args.split(",")) if x]:
            params.append(rust_type(a.split(":", 1)[1]))
        decls[name] = (rust_type(ret) if ret else "unit", params)
    return decls


def test_extern_block_matches_header():
    c, r = c_decls(), rust_decls()
    assert len(r) >= 10, "extern block not parsed"
    for name, (ret, params) in r.items():
        assert name in c, f"{name} is not declared in include/rnamc.h"
        cret, cparams = c[name]
        assert normalise(ret) == normalise(cret), f"{name}: return type {ret} vs C {cret}"
        assert len(params) == len(cparams), f"{name}: arity {len(params)} vs C {len(cparams)}"
        for k, (a, b) in enumerate(zip(params, cparams)):
            assert normalise(a) == normalise(b), f"{name}: parameter {k}: Rust {a} vs C {b}"


def test_durbin_extern_block_matches_header():
    c, r = c_decls(), rust_decls(DURBIN_SHIM)
    assert "rnamc_durbin_batch" in r
    for name, (ret, params) in r.items():
        assert name in c, f"{name} is not declared in include/rnamc.h"
        cret, cparams = c[name]
        assert normalise(ret) == normalise(cret), name
        assert len(params) == len(cparams), name
        for k, (a, b) in enumerate(zip(params, cparams)):
            assert normalise(a) == normalise(b), f"{name}: parameter {k}: Rust {a} vs C {b}"
    # the #[repr(C)] mirror of rnamc_align_scores: same fields, same order
    src = open(DURBIN_SHIM).read()
    body = re.search(r"pub struct AlignScoresC \{(.*?)\n\}", src, flags=re.S).group(1)
    rust_fields = [f.split(":")[0].strip() for f in body.split("\n") if ":" in f]
    hdr = re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)
    cbody = re.search(r"typedef struct rnamc_align_scores \{(.*?)\}", hdr, flags=re.S).group(1)
    c_fields = [re.sub(r"\[.*", "", f.split()[-1]) for f in cbody.split(";") if f.strip()]
    assert rust_fields == c_fields


def test_header_symbols_are_exported_and_bound():
    """every function of include/rnamc.h is exported by librnamc.so and known to the ctypes
    binding (and the other way round)"""
    from rna_algos_amd import _lib
    L = _lib.lib()
    names = set(c_decls())
    assert names == set(_lib.SYMBOLS), names ^ set(_lib.SYMBOLS)
    for n in names:
        assert hasattr(L, n), n


def test_batch_stats_struct_matches_header():
    """the ctypes mirror of rnamc_batch_stats has the header's fields in order (a binding of an
    older header is protected by rnamc_ctx_stats' size argument; this one must be current)"""
    from rna_algos_amd import _lib
    hdr = re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)
    body = re.search(r"typedef struct rnamc_batch_stats \{(.*?)\} rnamc_batch_stats;", hdr, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ty, names = decl.split(None, 1)
        for nm in names.split(","):
            fields.append((nm.strip(), ty))
    want = {"uint64_t": "c_ulong", "double": "c_double"}
    got = [(n, t.__name__) for n, t in _lib.BatchStats._fields_]
    assert got == [(n, want[t]) for n, t in fields]


def test_twoloop_struct_matches():
    src = open(SHIM).read()
    body = re.search(r"pub struct TwoloopScore \{(.*?)\}", src, flags=re.S).group(1)
    fields = [tuple(x.strip() for x in f.split(":")) for f in body.split(",") if f.strip()]
    assert fields == [("i", "u32"), ("j", "u32"), ("k", "u32"), ("l", "u32"), ("score", "f32")]
    hdr = re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)
    cbody = re.search(r"typedef struct rnamc_twoloop_score \{(.*?)\}", hdr, flags=re.S).group(1)
    assert re.sub(r"\s+", " ", cbody).strip() == "uint32_t i, j, k, l; float score;"


def test_shim_keeps_reference_semantics():
    """the three boundary bugs of round 1 stay fixed: no process-wide OnceLock keyed by the
    first set, FoldScores filled by default, panics carry librnamc's message"""
    src = open(SHIM).read()
    code = re.sub(r"//[^\n]*", "", src)
    assert "OnceLock" not in code
    assert "content_key(fold_score_sets)" in code and "rnamc_pool_set_params" in code
    # the batch entry drives a POOL: every visible device by default (null list, n_devices = 0), the
    # rank's own device under a one-process-per-GPU launch (LOCAL_RANK), or the RNAMC_DEVICES list
    assert "rnamc_bpp_batch_multi" in code and re.search(r"rnamc_pool_create\(p,\s*dev_ptr,\s*devices\.len\(\) as u32,", code)
    assert "std::ptr::null()" in code and '"LOCAL_RANK"' in code and '"RNAMC_DEVICES"' in code
    assert 'cfg!(feature = "no-fold-scores")' in code and 'feature = "fold-scores"' not in code
    assert "panic!()" not in code and "rnamc_last_error" in code
    assert "pub fn mccaskill_algo_batch<T>" in code


def test_dumper_field_table_reproduces_params_layout():
    """bindings/rust/dump_tables.rs writes rnamc_params field by field without linking
    librnamc: its table must give the library's offsets, counts and total size."""
    import ctypes as C
    from rna_algos_amd import _lib
    L = _lib.lib()
    src = open(DUMPER).read()
    body = src[src.index("FIELD ORDER BEGIN"):src.index("FIELD ORDER END")]
    off = 16  # abi_version, struct_bytes, table_id
    table = {}
    for line in body.splitlines():
        line = line.strip()
        m = re.match(r'b\.f32s\("([\w\.]+)",\s*(\d+),', line)
        if m:
            table[m.group(1)] = (off, int(m.group(2)))
            off += 4 * int(m.group(2))
        elif line.startswith("b.bytes.extend_from_slice(&seqs)"):
            off += 64 * 16
        elif line.startswith("b.bytes.extend_from_slice(&lens)"):
            off += 64
        elif re.match(r"b\.u32\(", line):
            off += 4
    total = (off + 7) // 8 * 8
    assert total == L.rnamc_params_sizeof()
    name, o, cnt = C.c_char_p(), C.c_uint64(), C.c_uint64()
    idx = 0
    seen = set()
    while L.rnamc_params_field(idx, C.byref(name), C.byref(o), C.byref(cnt)) == _lib.OK:
        key = name.value.decode()
        assert key in table, f"dumper lacks {key}"
        assert table[key] == (o.value, cnt.value), (key, table[key], o.value, cnt.value)
        seen.add(key)
        idx += 1
    assert seen == set(table)
    # and the non-float members sit where the header puts them
    from rna_algos_amd.utils import FoldScoreSets
    p = FoldScoreSets.synthetic(3)
    so, sc = table["turner.special_hairpin_scores"]
    tail = so + 4 * sc + 64 * 16 + 64
    n_special, min_len, max_ex, min_ex = p._buf[tail:tail + 16].view(np.uint32)
    assert n_special <= 64 and (min_len, max_ex, min_ex) == (3, 9, 10)
    assert table["contra.hairpin_scores_len"][0] == tail + 16
