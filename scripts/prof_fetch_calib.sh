# FETCH_SIZE calibration on known byte counts (guide: "calibrate on a known byte count in
# your own access pattern before trusting an absolute")
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/fetch_calib
rm -rf $OUT && mkdir -p $OUT
for P in 0 1 2; do
  echo "pattern $P: plain run"; timeout -k 5 60 $R/scripts/ubench/fetch_calib $P
  echo "pattern $P: counter run"
  timeout -k 5 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p$P -- $R/scripts/ubench/fetch_calib $P > $OUT/run$P.log 2>&1
  python3 - <<PY
import csv, glob
tot = 0.0
for f in glob.glob("$OUT/p$P/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and ("k_dword" in r["Kernel_Name"] or "k_lane" in r["Kernel_Name"] or "k_float4" in r["Kernel_Name"]):
            tot += float(r["Counter_Value"])
named = [l for l in open("$OUT/run$P.log") if "GB named" in l][0].strip()
gb = float(named.split(":")[1].split()[0])
print(f"{named}; FETCH_SIZE = {tot*1024/1e9:.3f} GB (KB units) -> counter/named = {tot*1024/1e9/gb:.3f}")
PY
done
