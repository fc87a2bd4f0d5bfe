# FETCH_SIZE of the sweep kernels with only some roles enabled (timing/traffic experiment)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ROLES in "$@"; do
  OUT=$R/gpurun_out/fetch_roles_$ROLES
  rm -rf $OUT && mkdir -p $OUT
  ROLES=$ROLES timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -- python3 $R/scripts/quick_timing.py top512 > $OUT/run.log 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "k_outside" if "k_outside" in r["Kernel_Name"] else "k_inside" if "k_inside" in r["Kernel_Name"] else "other"
        agg[k] += float(r["Counter_Value"])
print("ROLES=$ROLES", {k: round(v*1024*2/1e12, 3) for k, v in agg.items()}, "TB (FETCH_SIZE x2)")
PY
  grep -h rep0 $OUT/run.log
  find $OUT -name "*counter_collection.csv" -delete
done
