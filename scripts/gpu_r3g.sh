#!/bin/bash
# kernel trace of the banded tree sweep
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_band
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/scripts/tree_time.py 4096 0 3 tree_band=64 tree_short=1 > $OUT/traced.txt 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
# keep the trace of the last pass only, compact: name, start, end
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with open("$OUT/trace_compact.txt", "w") as o:
    for r in rows[-4400:]:
        kn = r["Kernel_Name"]
        short = "mid" if "k_tree_mid" in kn else ("in" + kn.split("<")[1].split(">")[0] if "inside2" in kn else ("out" + kn.split("<")[1].split(">")[0] if "outside2" in kn else kn[:30]))
        o.write(f"{short} {int(r['Start_Timestamp'])-t0} {int(r['End_Timestamp'])-int(r['Start_Timestamp'])} {r['Grid_Size_X'] if 'Grid_Size_X' in r else ''}\n")
PY
find $OUT/trace -name "*kernel_trace.csv" -delete
cat $OUT/traced.txt; cat $OUT/kernel_stats.csv | cut -c1-200
