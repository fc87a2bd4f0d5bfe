"""rna_algos_amd — MI355X-native McCaskill bpp hot path of heartsh/rna-algos.

Module names follow the reference crate (`rna_algos::{utils, mccaskill_algo,
centroid_fold}`).  The compute lives in librnamc.so (HIP, gfx950); importing a
submodule that needs it raises ImportError when the library is not built.
"""
from . import _lib  # noqa: F401

__all__ = ["utils", "mccaskill_algo", "centroid_fold"]
