# rocprofv3 kernel trace of one n=1024 Turner + CONTRAfold run (per-dispatch durations)
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_n1024
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/quick_timing.py n1024 > $OUT/run.log 2>&1
find $OUT -name "*.csv" | head
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof_n1024"
f = [p for p in glob.glob(out + "/**/*kernel_trace.csv", recursive=True)][0]
rows = list(csv.DictReader(open(f)))
print(len(rows), "dispatches")
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in by.items():
    print(k, len(v), "total_ms=%.1f" % (sum(v) / 1e6), "avg_us=%.1f" % (sum(v) / len(v) / 1e3))
# duration profile along the sweep for the first Turner inside pass
ins = [r for r in rows if "k_insideILb0" in r["Kernel_Name"]]
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in ins[:1021]]
print("inside turner per-launch us at d=4+[0,100,200,...]:", [round(durs[x], 1) for x in range(0, len(durs), 100)])
outs = [r for r in rows if "k_outsideILb0" in r["Kernel_Name"]]
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in outs[:1020]]
print("outside turner per-launch us (d descending):", [round(durs[x], 1) for x in range(0, len(durs), 100)])
gaps = [int(rows[x + 1]["Start_Timestamp"]) - int(rows[x]["End_Timestamp"]) for x in range(len(rows) - 1)]
gaps.sort()
print("median gap us", gaps[len(gaps) // 2] / 1e3)
PY
