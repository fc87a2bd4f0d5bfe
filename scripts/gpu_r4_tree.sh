set -e
cd $GRAFT_REPO_ROOT
for m in turner contra; do
  timeout -k 10 300 python bench.py --workload n4096 --model $m --steps 5 --warmup 1 > gpurun_out/r04_bench_n4096_$m.json 2>/dev/null
done
timeout -k 10 200 python bench.py --workload n1024 --model contra --steps 5 --warmup 1 > gpurun_out/r04_bench_n1024_contra.json 2>/dev/null
timeout -k 10 500 bash scripts/prof_tree.sh r04 4096 0 > gpurun_out/prof_tree_r04.log 2>&1
timeout -k 10 100 python scripts/tree_ms.py 1024 2048 4096 8192 16384 > gpurun_out/r04_tree_ms.txt 2>/dev/null
cat gpurun_out/r04_tree_ms.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04_pytest_c.log 2>&1; tail -3 gpurun_out/r04_pytest_c.log
timeout -k 10 600 python tests/soak_tree.py 30 405 > gpurun_out/r04_soak_tree.txt 2>&1; tail -1 gpurun_out/r04_soak_tree.txt
