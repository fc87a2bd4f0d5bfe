#!/bin/bash
# One GPU call that refreshes the round-3 records at HEAD (run on the GPU box via gpurun):
# GPU suite, soak, Turner batch bench + rocprofv3 kernel stats + PMC traffic, the single-sequence
# configs, the tree-order sweep's kernel stats / traffic.  (The driver's own command,
# `python bench.py --gpus 1 --steps 20 --warmup 5`, takes 9 minutes: a call of its own.)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout -k 10 600 python tests/soak.py 40 77 > gpurun_out/soak_r03.txt 2>&1; tail -1 gpurun_out/soak_r03.txt
timeout -k 10 700 bash scripts/prof_bench.sh r03 --steps 3 --warmup 3 > gpurun_out/prof_bench_r03.log 2>&1
timeout -k 10 400 bash scripts/prof_traffic.sh r03 --batch-count 1000 > gpurun_out/prof_traffic_r03.log 2>&1
for m in turner contra; do
  timeout -k 10 300 python bench.py --workload n4096 --model $m --steps 5 --warmup 1 > gpurun_out/bench_n4096_$m.json 2>/dev/null
done
timeout -k 10 200 python bench.py --workload n1024 --model contra --steps 5 --warmup 1 > gpurun_out/bench_n1024_contra.json 2>/dev/null
timeout -k 10 500 bash scripts/prof_tree.sh r03 4096 0 > gpurun_out/prof_tree_r03.log 2>&1
