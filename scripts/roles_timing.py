"""Ad-hoc: time the outside latency kernels with single roles switched off (debug build,
RNAMC_LIB=.../librnamc_dbg.so).  The first call runs everything, so later calls with a role
off read the (valid) values the first one left in the workspace."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import quick_timing as Q  # noqa: E402
from rna_algos_amd import workloads as WL0  # noqa: E402
from rna_algos_amd.utils import FoldScoreSets  # noqa: E402
from rna_algos_amd.mccaskill_algo import Context  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = Context(FoldScoreSets.synthetic(1), device=0)
ctx.set("profile", 2)
seq = [WL0.synthetic_seq(n, n)]
for contra in (False, True):
    for roles, name in ((15, "all"), (11, "no mb"), (47, "no tail"), (31, "no head"), (14, "outside only")):
        ctx.set("debug_roles", roles)
        Q.run(ctx, seq, contra, reps=1, label=f"n{n} [{name}]")
