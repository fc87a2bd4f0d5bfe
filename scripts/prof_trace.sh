# usage: bash scripts/prof_trace.sh <quick_timing workload>  -> per-dispatch durations along the sweep
set -e
cd /tmp && export TMPDIR=/tmp
W=${1:-top512}
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$W
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/quick_timing.py $W > $OUT/run.log 2>&1
grep -h "rep0" $OUT/run.log
python3 - <<PY
import csv, glob, os
out = "$OUT"
f = sorted(glob.glob(out + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for name in ["k_inside", "k_outside"]:
    ins = [r for r in rows if name in r["Kernel_Name"]]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in ins]
    print(name, len(ins), "total ms %.1f" % (sum(d) / 1e3))
    print("  us every 64th launch:", [round(d[x], 1) for x in range(0, len(d), 64)])
with open(out + "/durations.txt", "w") as fh:
    for name in ["k_inside", "k_outside"]:
        ins = [r for r in rows if name in r["Kernel_Name"]]
        fh.write(name + " " + " ".join(str((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) // 100 / 10) for r in ins) + "\n")
gaps = sorted(int(rows[x + 1]["Start_Timestamp"]) - int(rows[x]["End_Timestamp"]) for x in range(len(rows) - 1))
print("gap us: median %.2f p90 %.2f total ms %.1f" % (gaps[len(gaps)//2]/1e3, gaps[int(len(gaps)*0.9)]/1e3, sum(gaps)/1e6))
PY
cp $(find $OUT -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
