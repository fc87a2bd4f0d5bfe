# PMC counter passes (separate runs, no tracing domains) over a small batch.
# usage: bash scripts/prof_pmc.sh <workload args for quick_timing.py>
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
rm -rf $OUT && mkdir -p $OUT
W="${@:-batch128}"
export CONTRA=0 GSIZES=1024
i=0
for SET in \
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
 "FETCH_SIZE" \
 "WRITE_SIZE" ; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/scripts/quick_timing.py $W > $OUT/run$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/run$i.log; }
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:28s} {x:.4g}")
PY
# keep only the summary + small logs (the per-dispatch CSVs are large)
find $OUT -name "*counter_collection.csv" -size +8M -delete
