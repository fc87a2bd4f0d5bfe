# PMC counter passes of the tree-order mode on a slice of the bench batch (separate runs, no tracing
# domains), aggregated per kernel.  usage: bash scripts/prof_pmc_tree_batch.sh <tag> [count] [stride]
set -e
cd /tmp && export TMPDIR=/tmp
TAG="${1:-r04}"; CNT=${2:-250}; export TREE_STRIDE=${3:-40}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_tree_batch_$TAG
rm -rf $OUT && mkdir -p $OUT
i=0
for SET in \
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA" \
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
 "FETCH_SIZE" "WRITE_SIZE" \
; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/scripts/tree_batch_knobs.py $CNT 0 0 > $OUT/run$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/run$i.log; }
done
tail -1 $OUT/run1.log
python3 - <<PY
import csv, glob, os, collections
out = "$OUT"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
names = ("k_tlane_gen", "k_tlane_inside", "k_tlane_outside", "k_tree_mid_mx", "k_tree_mid", "k_tree_static", "k_tree_init", "k_tlane_list",
         "k_tlane_spread", "k_tree_ext", "k_tree_finalize")
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        k = next((x for x in names if x in kn), "other")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVES", "FETCH_SIZE"):
            agg[k]["launches(" + r["Counter_Name"] + " pass)"] += 1
with open(out + "/summary.txt", "w") as fh:
    for k, v in agg.items():
        print(k); fh.write(k + "\n")
        for c, x in sorted(v.items()):
            line = f"   {c:32s} {x:.5g}"
            print(line); fh.write(line + "\n")
PY
find $OUT -name "*counter_collection.csv" -delete
