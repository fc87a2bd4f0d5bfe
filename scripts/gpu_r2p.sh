set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( while true; do sleep 60; echo "[alive $(date +%T)]"; done ) &
KEEP=$!
trap "kill $KEEP 2>/dev/null" EXIT
timeout -k 10 200 python bench.py --batch-count 600 --steps 3 --warmup 5 --time-budget-s 28 --no-cpu-baseline --no-n4096 > gpurun_out/bench_budget_small.json 2> gpurun_out/bench_budget_small.err || { tail -5 gpurun_out/bench_budget_small.err; exit 1; }
python - <<PY
import json
r=json.loads(open("gpurun_out/bench_budget_small.json").read().strip().splitlines()[-1])
print("small:", r["steps"], r.get("steps_requested"), r["warmup"], r.get("warmup_requested"), r.get("notes"), r["value_with_transfers"], r["wall_s"])
PY
# the driver's own command, end to end
START=$(date +%s)
timeout -k 10 640 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_rehearsal.json 2> gpurun_out/bench_driver_rehearsal.err || { tail -5 gpurun_out/bench_driver_rehearsal.err; exit 1; }
END=$(date +%s)
echo "driver command took $((END-START)) s"
python - <<PY
import json
r=json.loads(open("gpurun_out/bench_driver_rehearsal.json").read().strip().splitlines()[-1])
print("full:", r["value"], r["ms_per_step"], r["steps"], r.get("steps_requested"), r["warmup"], r.get("warmup_requested"), r.get("notes"), r["value_with_transfers"], r.get("ms_per_seq_n4096"), (r.get("cpu_baseline") or {}).get("value"), r["wall_s"], r["parity_check"][:20])
PY
