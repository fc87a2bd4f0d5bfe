"""ctypes binding of librnamc.so (the C ABI declared in include/rnamc.h).

The HIP extension is the product: if the shared library is missing this module
raises ImportError("... not built ...") — there is no CPU fallback and nothing
under oracle/ is ever imported from here.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RNAMC_LIB") or os.path.join(_HERE, "librnamc.so")

OK = 0
ERR_INVALID_ARG, ERR_INVALID_BASE, ERR_EMPTY_SEQ, ERR_SEQ_TOO_LONG = 1, 2, 3, 4
ERR_NO_DEVICE, ERR_OOM, ERR_HIP, ERR_IO, ERR_FORMAT = 5, 6, 7, 8, 9

# every entry point include/rnamc.h declares
SYMBOLS = [
    "rnamc_abi_version", "rnamc_params_sizeof", "rnamc_strerror", "rnamc_last_error",
    "rnamc_bytes2seq", "rnamc_bpp_len", "rnamc_bpp_index",
    "rnamc_fold_score_sets_new", "rnamc_fold_score_sets_accumulate",
    "rnamc_fold_score_sets_transfer", "rnamc_params_new", "rnamc_params_synthetic",
    "rnamc_params_save", "rnamc_params_load", "rnamc_params_field",
    "rnamc_params_set_special_hairpins", "rnamc_params_set_hairpin_limits",
    "rnamc_ctx_create", "rnamc_ctx_destroy", "rnamc_ctx_set", "rnamc_ctx_set_params",
    "rnamc_bpp_batch", "rnamc_bpp_batch_device", "rnamc_ctx_last_stats", "rnamc_ctx_stats",
    "rnamc_debug_fetch", "rnamc_fold_scores", "rnamc_fold_sums", "rnamc_centroid_fold",
    "rnamc_centroid_fold_multi",
    "rnamc_align_scores_new", "rnamc_align_scores_transfer", "rnamc_durbin_batch",
    "rnamc_pool_create", "rnamc_pool_destroy", "rnamc_pool_size", "rnamc_pool_ctx",
    "rnamc_pool_set_params", "rnamc_pool_set", "rnamc_bpp_batch_multi", "rnamc_shard_plan",
    "rnamc_sweep_cost",
]


class RnamcError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = lib().rnamc_strerror(status).decode()
        if detail:
            msg += ": " + detail
        super().__init__(f"rnamc status {status}: {msg}")


class BatchStats(C.Structure):
    _fields_ = [
        ("n_groups", C.c_uint64), ("launches_inside", C.c_uint64),
        ("launches_outside", C.c_uint64), ("launches_other", C.c_uint64),
        ("ms_inside", C.c_double), ("ms_outside", C.c_double), ("ms_other", C.c_double),
        ("workspace_bytes", C.c_uint64),
        ("launches_outside_main", C.c_uint64), ("launches_outside_tail", C.c_uint64),
        ("launches_outside_small", C.c_uint64),
        ("ms_outside_main", C.c_double), ("ms_outside_tail", C.c_double),
        ("ms_outside_small", C.c_double), ("tree_side_stream", C.c_uint64),
    ]


_lib = None


def _preload_torch_hip():
    """One HIP runtime per process.  torch bundles its own libamdhip64.so / libhsa-runtime64.so;
    librnamc.so needs `libamdhip64.so.7`.  If librnamc is loaded first it brings in /opt/rocm's
    runtime, and a later `import torch` adds its bundled pair as a second HSA runtime, which
    then sees no GPU ("No HIP GPUs are available").  Loading torch's copy first (without
    importing torch) lets the dynamic loader satisfy librnamc's NEEDED entry with it (same
    SONAME) and torch find its own library already mapped.  RNAMC_NO_TORCH_PRELOAD=1 skips this
    (processes that never use torch and want /opt/rocm's runtime)."""
    import importlib.util
    import sys
    if os.environ.get("RNAMC_NO_TORCH_PRELOAD") == "1" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    _preload_torch_hip()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). rna_algos_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, u8p, u32p, u64p, f32p = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), \
        C.POINTER(C.c_uint64), C.POINTER(C.c_float)
    L.rnamc_abi_version.restype = C.c_uint32
    L.rnamc_params_sizeof.restype = C.c_size_t
    L.rnamc_strerror.restype = C.c_char_p
    L.rnamc_strerror.argtypes = [C.c_int]
    L.rnamc_last_error.restype = C.c_char_p
    L.rnamc_bytes2seq.argtypes = [vp, C.c_uint64, vp]
    L.rnamc_bpp_len.restype = C.c_uint64
    L.rnamc_bpp_len.argtypes = [C.c_uint32]
    L.rnamc_bpp_index.restype = C.c_uint64
    L.rnamc_bpp_index.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    L.rnamc_fold_score_sets_new.argtypes = [C.c_float, vp]
    L.rnamc_fold_score_sets_accumulate.argtypes = [vp]
    L.rnamc_fold_score_sets_transfer.argtypes = [vp, vp]
    L.rnamc_params_new.argtypes = [C.c_float, vp]
    L.rnamc_params_synthetic.argtypes = [C.c_uint64, vp]
    L.rnamc_params_save.argtypes = [vp, C.c_char_p]
    L.rnamc_params_load.argtypes = [C.c_char_p, vp]
    L.rnamc_params_field.argtypes = [C.c_uint32, C.POINTER(C.c_char_p), u64p, u64p]
    L.rnamc_params_set_special_hairpins.argtypes = [vp, C.c_uint32, vp, vp, vp]
    L.rnamc_params_set_hairpin_limits.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32]
    L.rnamc_ctx_create.argtypes = [vp, C.c_int, C.c_uint64, C.POINTER(vp)]
    L.rnamc_ctx_destroy.argtypes = [vp]
    L.rnamc_ctx_destroy.restype = None
    L.rnamc_ctx_set.argtypes = [vp, C.c_char_p, C.c_int64]
    L.rnamc_ctx_set_params.argtypes = [vp, vp]
    L.rnamc_bpp_batch.argtypes = [vp, C.c_uint32, vp, vp, C.c_int, C.c_int, vp, vp, vp]
    L.rnamc_bpp_batch_device.argtypes = [vp, C.c_uint32, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]
    L.rnamc_ctx_last_stats.argtypes = [vp, C.POINTER(BatchStats)]
    L.rnamc_ctx_stats.argtypes = [vp, vp, C.c_uint64, u64p]
    L.rnamc_pool_create.argtypes = [vp, vp, C.c_uint32, C.c_uint64, C.POINTER(vp)]
    L.rnamc_pool_destroy.argtypes = [vp]
    L.rnamc_pool_destroy.restype = None
    L.rnamc_pool_size.argtypes = [vp]
    L.rnamc_pool_size.restype = C.c_uint32
    L.rnamc_pool_ctx.argtypes = [vp, C.c_uint32]
    L.rnamc_pool_ctx.restype = vp
    L.rnamc_pool_set_params.argtypes = [vp, vp]
    L.rnamc_pool_set.argtypes = [vp, C.c_char_p, C.c_int64]
    L.rnamc_bpp_batch_multi.argtypes = [vp, C.c_uint32, vp, vp, C.c_int, C.c_int, vp, vp, vp]
    L.rnamc_shard_plan.argtypes = [C.c_uint32, vp, C.c_uint32, vp]
    L.rnamc_sweep_cost.argtypes = [C.c_uint32, vp, vp]
    L.rnamc_debug_fetch.argtypes = [vp, C.c_uint32, C.c_int, vp]
    L.rnamc_fold_scores.argtypes = [vp, vp, C.c_uint32, C.c_int, C.c_int, vp, vp, vp, vp,
                                    C.c_uint64, u64p]
    L.rnamc_centroid_fold.argtypes = [vp, C.c_uint32, C.c_float, vp, C.c_uint32, u32p, f32p]
    L.rnamc_fold_sums.argtypes = [vp, vp, C.c_uint32, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    L.rnamc_centroid_fold_multi.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, vp, vp]
    L.rnamc_align_scores_new.argtypes = [C.c_float, vp]
    L.rnamc_align_scores_transfer.argtypes = [vp]
    L.rnamc_durbin_batch.argtypes = [vp, vp, C.c_uint32, vp, vp, C.c_uint32, vp, vp, vp, vp]
    _lib = L
    return L


def check(status):
    if status != OK:
        raise RnamcError(status, lib().rnamc_last_error().decode(errors="replace"))
