# HBM traffic of the tree-order BATCH form: FETCH_SIZE / WRITE_SIZE in PMC passes of their own over one pass of a
# 1 000-sequence slice of the bench batch (every 10th sequence), the slice's pass time outside the profiler, and the
# kernel stats of the same slice.  usage: bash scripts/prof_traffic_tree_batch.sh <tag> [contra 0|1]
set -e
TAG=${1:-r04}; CONTRA=${2:-0}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/traffic_tree_batch_$TAG
rm -rf $OUT && mkdir -p $OUT
export TREE_STRIDE=10
python3 $R/scripts/tree_batch_knobs.py 1000 $CONTRA 2 > $OUT/plain.txt 2>&1 || { tail -5 $OUT/plain.txt; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/scripts/tree_batch_knobs.py 1000 $CONTRA 0 > $OUT/traced.txt 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
find $OUT/trace -name "*kernel_trace.csv" -delete
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $R/scripts/tree_batch_knobs.py 1000 $CONTRA 0 > $OUT/pmc_$C.txt 2> $OUT/$C.err || { tail -5 $OUT/$C.err; exit 1; }
done
python3 - <<PY
import csv, glob, json, collections, re
out = "$OUT"
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
names = ("k_tlane_gen", "k_tlane_inside", "k_tlane_outside", "k_tree_mid_mx", "k_tree_static", "k_tree_init", "k_tlane_list",
         "k_tlane_spread", "k_tree_ext", "k_tree_finalize")
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        k = next((x for x in names if x in kn), "other")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "FETCH_SIZE": n[k] += 1
plain = open(out + "/plain.txt").read().strip().splitlines()
m = re.search(r"pass\s+([0-9.]+) ms", plain[-1]); nt = int(re.search(r"(\d+) nt", plain[-2]).group(1))
res = {"workload": "tree-order batch form, every 10th sequence of the 10k bench batch (1 000 sequences), contra=$CONTRA, one pass per PMC run",
       "nt": nt, "pass_ms": float(m.group(1)), "launches": dict(n)}
tot = 0.0
for k in names + ("other",):
    f, w = agg[k]["FETCH_SIZE"] * 1024, agg[k]["WRITE_SIZE"] * 1024
    if f == 0 and w == 0: continue
    # gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X guide, HBM section): doubled
    res[k] = {"fetch_bytes_raw": f, "fetch_bytes_x2": 2 * f, "write_bytes": w, "launches": n[k]}
    tot += 2 * f + w
res["total_bytes_x2"] = tot
res["achieved_GBps"] = tot / (res["pass_ms"] * 1e-3) / 1e9
res["plain_run"] = plain[-2:]
json.dump(res, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find $OUT -name "*counter_collection.csv" -delete
head -12 $OUT/kernel_stats.csv
