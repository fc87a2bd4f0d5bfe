set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "--group-ws-gb 128 --group-max-nt 4194304" "--group-ws-gb 180 --group-max-nt 8388608" "--group-ws-gb 64 --group-max-nt 1500000"; do
  timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-n4096 --no-transfers --no-kernel-timing $cfg > gpurun_out/bench_grp.json 2> gpurun_out/bench_grp.err || { tail -5 gpurun_out/bench_grp.err; exit 1; }
  python - <<PY
import json
r=json.loads(open("gpurun_out/bench_grp.json").read().strip().splitlines()[-1])
print("$cfg", round(r["value"]), round(r["ms_per_step"]), "inside", round(r["roofline_inside"]["ms_per_step"]), "outside", round(r["roofline_outside_sweep"]["ms_per_step"]))
PY
done
