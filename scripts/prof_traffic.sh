# HBM traffic of the sweep kernels from PMC counters, separate passes (no tracing domains):
# FETCH_SIZE, WRITE_SIZE (KB at the L2's fabric side), per kernel, on a reduced batch.
# usage: bash scripts/prof_traffic.sh <tag> [bench args]
set -e
TAG=${1:-r01}; shift || true
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_traffic_$TAG
rm -rf $OUT && mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $R/bench.py "$@" --no-cpu-baseline --no-kernel-timing --no-n4096 --steps 1 --warmup 0 > $OUT/bench_$C.json 2> $OUT/$C.err || { tail -5 $OUT/$C.err; exit 1; }
done
python3 - <<PY
import csv, glob, json, collections
out = "$OUT"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "k_outside" in kn:
            k = "k_outside_main" if ", 5>" in kn else "k_outside_tail" if ", 2>" in kn else "k_outside_small"
        elif "k_tree_inside" in kn:
            k = "k_tree_inside"
        elif "k_tree_outside" in kn:
            k = "k_tree_outside"
        elif "k_inside" in kn or "k_pair_tail" in kn:
            k = "k_inside"
        else:
            k = "other"
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "FETCH_SIZE": n[k] += 1
b = json.loads(open(out + "/bench_FETCH_SIZE.json").read().strip().splitlines()[-1])
res = {"bench_config": b["config"], "launches": dict(n), "counters_KB": {k: dict(v) for k, v in agg.items()}}
for k in ("k_inside", "k_outside_main", "k_outside_tail", "k_outside_small", "k_tree_inside", "k_tree_outside"):
    if k not in agg: continue
    f, w = agg[k]["FETCH_SIZE"] * 1024, agg[k]["WRITE_SIZE"] * 1024
    # gfx950: FETCH_SIZE tallies 128-B requests at 64 B (guide §HBM); calibrated on our own
    # access patterns in profiles/r01_fetch_size_calibration.txt: 0.50 x bytes for all of them
    res[k] = {"fetch_bytes_raw": f, "fetch_bytes_x2": 2 * f, "write_bytes": w, "launches": n[k],
              "per_launch_raw": (f + w) / max(n[k], 1), "per_launch_x2": (2 * f + w) / max(n[k], 1)}
for key in ("roofline", "roofline_tail", "roofline_outside_sweep", "roofline_inside", "tree", "ms_per_seq", "ms_per_seq_tree"):
    if key in b: res[key] = b[key]
json.dump(res, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
find $OUT -name "*counter_collection.csv" -delete
