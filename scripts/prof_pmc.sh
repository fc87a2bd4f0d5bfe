# PMC counter passes (separate runs, no tracing domains), aggregated per kernel.
# usage: bash scripts/prof_pmc.sh <quick_timing workload>
set -e
cd /tmp && export TMPDIR=/tmp
W="${1:-top256}"
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$W
rm -rf $OUT && mkdir -p $OUT
export CONTRA=0 GSIZES=${GSIZES:-1024}
i=0
for SET in \
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA" \
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/scripts/quick_timing.py $W > $OUT/run$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/run$i.log; }
done
grep -h "rep0" $OUT/run1.log
python3 - <<PY
import csv, glob, os, collections
out = "$OUT"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "k_tree" in kn:
            k = next(x for x in ("k_tree_inside", "k_tree_outside", "k_tree_mid", "k_tree_ext", "k_tree") if x in kn)
        elif "k_outside" in kn:
            k = "k_outside_main" if ", 5>" in kn else "k_outside_tail" if ", 2>" in kn else "k_outside_small"
        elif "k_inside2" in kn:
            k = "k_inside2"
        elif "k_inside" in kn or "k_pair_tail" in kn:
            k = "k_inside_small"
        else:
            k = "other"
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open(out + "/summary.txt", "w") as fh:
    for k, v in agg.items():
        print(k); fh.write(k + "\n")
        for c, x in sorted(v.items()):
            line = f"   {c:32s} {x:.5g}"
            print(line); fh.write(line + "\n")
PY
find $OUT -name "*counter_collection.csv" -delete
