set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_big
rm -rf $OUT && mkdir -p $OUT
export CONTRA=0 GSIZES=1024
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1 -- python3 $GRAFT_REPO_ROOT/scripts/quick_timing.py batch1024 > $OUT/run1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/p2 -- python3 $GRAFT_REPO_ROOT/scripts/quick_timing.py batch1024 > $OUT/run2.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_big"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-30:] + ("<outside>" if "k_outside" in r["Kernel_Name"] else "<inside>" if "k_inside" in r["Kernel_Name"] else "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:28s} {x:.4g}")
PY
grep -h batch $OUT/run1.log $OUT/run2.log
find $OUT -name "*counter_collection.csv" -delete
