# rocprofv3 summaries of the tree-order mode on one sequence (run on the GPU box via gpurun):
# kernel trace + stats, then FETCH_SIZE / WRITE_SIZE in passes of their own.
# usage: bash scripts/prof_tree.sh <tag> [n] [contra 0|1]
set -e
TAG=${1:-r03}; N=${2:-4096}; CONTRA=${3:-0}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_tree_$TAG
rm -rf $OUT && mkdir -p $OUT
python3 $R/scripts/tree_time.py $N $CONTRA 5 > $OUT/plain.txt 2>&1 || { tail -5 $OUT/plain.txt; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/scripts/tree_time.py $N $CONTRA 3 > $OUT/traced.txt 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
find $OUT/trace -name "*kernel_trace.csv" -delete
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $R/scripts/tree_time.py $N $CONTRA 1 > $OUT/pmc_$C.txt 2> $OUT/$C.err || { tail -5 $OUT/$C.err; exit 1; }
done
python3 - <<PY
import csv, glob, json, collections
out = "$OUT"
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        k = ("k_tree_inside" if "k_tree_inside" in kn else "k_tree_outside" if "k_tree_outside" in kn
             else "k_tree_mid" if "k_tree_mid" in kn else "k_tree_ext" if "k_tree_ext" in kn else "other")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "FETCH_SIZE": n[k] += 1
res = {"workload": "one synthetic sequence n=$N, contra=$CONTRA, tree-order mode, one call per PMC pass",
       "launches": dict(n)}
tot = 0.0
for k in ("k_tree_inside", "k_tree_outside", "k_tree_mid", "k_tree_ext"):
    f, w = agg[k]["FETCH_SIZE"] * 1024, agg[k]["WRITE_SIZE"] * 1024
    # gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X guide, HBM section): doubled
    res[k] = {"fetch_bytes_raw": f, "fetch_bytes_x2": 2 * f, "write_bytes": w, "launches": n[k],
              "per_launch_x2": (2 * f + w) / max(n[k], 1)}
    tot += 2 * f + w
res["total_bytes_x2"] = tot
res["plain_run"] = open(out + "/plain.txt").read().strip().splitlines()[-3:]
json.dump(res, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find $OUT -name "*counter_collection.csv" -delete
head -12 $OUT/kernel_stats.csv
