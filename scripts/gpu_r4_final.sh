#!/bin/bash
# One GPU call that refreshes the round-4 records at HEAD (run on the GPU box via gpurun):
# Turner batch bench + rocprofv3 kernel stats + PMC traffic, the single-sequence configs, the
# tree-order sweep's kernel stats / traffic, the soaks.  (The GPU suite and the driver's own command,
# `python bench.py --gpus 1 --steps 20 --warmup 5` — 9 minutes — are calls of their own.)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 700 bash scripts/prof_bench.sh r04 --steps 3 --warmup 3 > gpurun_out/prof_bench_r04.log 2>&1
timeout -k 10 400 bash scripts/prof_traffic.sh r04 --batch-count 1000 > gpurun_out/prof_traffic_r04.log 2>&1
for m in turner contra; do
  timeout -k 10 300 python bench.py --workload n4096 --model $m --steps 5 --warmup 1 > gpurun_out/r04_bench_n4096_$m.json 2>/dev/null
done
timeout -k 10 200 python bench.py --workload n1024 --model contra --steps 5 --warmup 1 > gpurun_out/r04_bench_n1024_contra.json 2>/dev/null
timeout -k 10 500 bash scripts/prof_tree.sh r04 4096 0 > gpurun_out/prof_tree_r04.log 2>&1
timeout -k 10 600 python tests/soak.py 40 404 > gpurun_out/r04_soak.txt 2>&1; tail -1 gpurun_out/r04_soak.txt
timeout -k 10 600 python tests/soak_tree.py 30 404 > gpurun_out/r04_soak_tree.txt 2>&1; tail -1 gpurun_out/r04_soak_tree.txt
