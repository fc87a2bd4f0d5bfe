#!/bin/bash
# after the eager side-stream fix: GPU suite, then the round's Turner batch records
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1 || { tail -20 gpurun_out/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/pytest_gpu.log
timeout -k 10 700 bash scripts/prof_bench.sh r03 --steps 3 --warmup 3 > gpurun_out/prof_bench_r03.log 2>&1 || { tail -20 gpurun_out/prof_bench_r03.log; exit 1; }
python3 - <<PY
import json
d = json.loads(open("gpurun_out/prof_bench_r03/bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "value_with_transfers", "ms_per_seq_n4096", "ms_per_seq_n4096_tree", "parity_check", "wall_s")})
PY
