set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( while true; do sleep 60; echo "[alive $(date +%T)]"; done ) &
KEEP=$!
trap "kill $KEEP 2>/dev/null" EXIT
timeout -k 10 900 python tests/soak.py 60 4242 > gpurun_out/soak_r02.log 2>&1 || { tail -5 gpurun_out/soak_r02.log; exit 1; }
tail -3 gpurun_out/soak_r02.log
grep -c "ok " gpurun_out/soak_r02.log
timeout -k 10 400 python bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-n4096 > gpurun_out/bench_w3.json 2> gpurun_out/bench_w3.err || { tail -5 gpurun_out/bench_w3.err; exit 1; }
python - <<PY
import json
r=json.loads(open("gpurun_out/bench_w3.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r["with_transfers"], r.get("notes"), r["wall_s"])
PY
