# round 3, GPU call A: new tests, tree bench line + profiles
set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r3a.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu_r3a.log
python bench.py --workload n4096 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_n4096_turner.json 2> gpurun_out/r03_bench_n4096_turner.err; tail -c 600 gpurun_out/r03_bench_n4096_turner.json
python bench.py --workload n1024 --model contra --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_n1024_contra.json 2> gpurun_out/r03_bench_n1024_contra.err; tail -c 300 gpurun_out/r03_bench_n1024_contra.json
bash scripts/prof_tree.sh r03_n4096 4096 0 > gpurun_out/prof_tree_r03.log 2>&1; tail -30 gpurun_out/prof_tree_r03.log
