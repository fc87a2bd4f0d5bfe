"""Does running two lock-step groups at once (the inside sweep of one beside the outside sweep
of the other) buy throughput?  Two contexts on one GPU, each fed half of the batch from its own
host thread, against one context with the whole batch."""
import sys, os, time, threading
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rna_algos_amd import workloads as W
from rna_algos_amd.utils import FoldScoreSets
from rna_algos_amd.mccaskill_algo import Context

count = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
lens = W.batch_lengths(10000)
order = np.argsort(-lens, kind="stable")[:count]
seqs = [W.synthetic_seq(int(lens[i]), (10000 << 32) + int(i)) for i in order]
P = FoldScoreSets.synthetic(1)
dev = torch.device("cuda:0")


def prep(ss):
    ln = np.array([len(s) for s in ss], dtype=np.uint64)
    off = np.zeros(len(ss) + 1, dtype=np.uint64); np.cumsum(ln, out=off[1:])
    oo = np.zeros(len(ss) + 1, dtype=np.uint64); np.cumsum(ln * (ln + np.uint64(1)) // np.uint64(2), out=oo[1:])
    b = torch.from_numpy(np.concatenate(ss)).to(dev)
    o = torch.empty(int(oo[-1]), dtype=torch.float32, device=dev)
    z = torch.empty(len(ss), dtype=torch.float32, device=dev)
    return ss, off, oo, b, o, z


def run(ctx, pk, stream):
    ss, off, oo, b, o, z = pk
    ctx.bpp_batch_device(len(ss), b.data_ptr(), off, False, False, o.data_ptr(), oo, z.data_ptr(), stream)


gb = int(os.environ.get("WSGB", "64"))
one = Context(P, device=0); one.set("group_ws_bytes", gb << 30)
whole = prep(seqs)
st0 = torch.cuda.Stream()
for rep in range(2):
    torch.cuda.synchronize(); t = time.time()
    run(one, whole, st0.cuda_stream); torch.cuda.synchronize()
    print(f"one context, {count} sequences: {time.time() - t:.2f} s", flush=True)
one.close(); del whole; torch.cuda.empty_cache()
# halves with the same length mix: groups of the two contexts are offset by half a group
halves = [prep(seqs[0::2]), prep(seqs[1::2])]
ctxs = [Context(P, device=0) for _ in range(2)]
for c in ctxs:
    c.set("group_ws_bytes", (gb // 2) << 30)
    c.set("group_max_nt", 1 << 20)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for rep in range(2):
    torch.cuda.synchronize(); t = time.time()
    th = [threading.Thread(target=run, args=(ctxs[k], halves[k], streams[k].cuda_stream)) for k in range(2)]
    # stagger: the second context starts when the first is about in its outside sweep
    th[0].start(); time.sleep(float(os.environ.get("STAGGER", "0.0"))); th[1].start()
    for x in th: x.join()
    torch.cuda.synchronize()
    print(f"two contexts, half each (half-size groups): {time.time() - t:.2f} s", flush=True)
