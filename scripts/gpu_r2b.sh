# round 2, second call: k_head (LDS-staged probe windows) parity + timing against the gather form
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lds_staged or two_kernel or trnas or bench_scale or random_small" 2>&1 | tee gpurun_out/pytest_gpu_b.log | tail -15
for H in 0 1; do
  timeout -k 10 300 python bench.py --steps 1 --warmup 1 --batch-count 2000 --no-cpu-baseline --no-n4096 --set head_lds=$H > gpurun_out/bench_head$H.json 2> gpurun_out/bench_head$H.err || { tail -5 gpurun_out/bench_head$H.err; exit 1; }
  python - <<PY
import json
r=json.load(open("gpurun_out/bench_head$H.json"))
print("head_lds=$H", r["value"], r["ms_per_step"], "inside", r["roofline_inside"]["ms_per_step"], "outside", r["roofline_outside_sweep"]["ms_per_step"], "main", r["roofline"]["avg_launch_ms"], "tail", r["roofline_tail"]["avg_launch_ms"], "head", r.get("roofline_head",{}).get("avg_launch_ms"))
PY
done
