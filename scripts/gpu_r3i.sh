# round 3, banded tree sweep: full GPU suite, n = 4096 / 1024 bench lines, kernel stats + PMC traffic
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r3i.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/pytest_gpu_r3i.log
[ $rc -eq 0 ] || exit 1
python bench.py --workload n4096 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r03b_bench_n4096_turner.json 2> gpurun_out/r03b_bench_n4096_turner.err && tail -c 400 gpurun_out/r03b_bench_n4096_turner.json &&
python bench.py --workload n4096 --model contra --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r03b_bench_n4096_contra.json 2> gpurun_out/r03b_bench_n4096_contra.err && tail -c 300 gpurun_out/r03b_bench_n4096_contra.json &&
python bench.py --workload n1024 --model contra --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r03b_bench_n1024_contra.json 2> gpurun_out/r03b_bench_n1024_contra.err && tail -c 300 gpurun_out/r03b_bench_n1024_contra.json &&
bash scripts/prof_tree.sh r03b_n4096 4096 0 > gpurun_out/prof_tree_r03b.log 2>&1; tail -12 gpurun_out/prof_tree_r03b.log
