"""Which HIP / HSA runtimes end up mapped, and does the second initialiser still see the GPU?
argv[1]: 'rnamc_first' or 'torch_first'.  (Diagnosis of the round-2 'No HIP GPUs are available'.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

def maps():
    out = set()
    for line in open("/proc/self/maps"):
        p = line.split()[-1]
        if any(k in p for k in ("amdhip64", "hsa-runtime", "libhsakmt", "librnamc.so")):
            out.add(p)
    return sorted(out)

def use_rnamc():
    from rna_algos_amd.utils import FoldScoreSets
    from rna_algos_amd.mccaskill_algo import Context
    c = Context(FoldScoreSets.synthetic(1), device=0)
    m, z = c.bpp_batch([np.arange(60, dtype=np.uint8) % 4], False, False)
    c.close()
    return float(z[0])

def use_torch():
    import torch
    ok = torch.cuda.is_available()
    x = torch.ones(4, device="cuda:0") if ok else None
    return ok, (float(x.sum()) if ok else None)

order = sys.argv[1]
print("order:", order, "HIP_VISIBLE_DEVICES=%s ROCR_VISIBLE_DEVICES=%s" % (os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES")))
try:
    if order == "rnamc_first":
        print("rnamc:", use_rnamc()); print("  mapped:", maps())
        print("torch:", use_torch()); print("  mapped:", maps())
        print("rnamc again:", use_rnamc())
    else:
        print("torch:", use_torch()); print("  mapped:", maps())
        print("rnamc:", use_rnamc()); print("  mapped:", maps())
        print("torch again:", use_torch())
except Exception as e:
    print("FAILED:", type(e).__name__, e); print("  mapped:", maps())
