# round 2, first call: tests, ubench of the speculative 8-lane logsumexp, the new bench flow
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/ubench/build.sh
timeout -k 10 120 ./scripts/ubench/lse_latency > gpurun_out/r02_lse_latency.txt 2>&1 && cat gpurun_out/r02_lse_latency.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tee gpurun_out/pytest_gpu.log | tail -5
timeout -k 10 600 python bench.py --steps 2 --warmup 2 > gpurun_out/bench_r2a.json 2> gpurun_out/bench_r2a.err || { tail -20 gpurun_out/bench_r2a.err; exit 1; }
cat gpurun_out/bench_r2a.json
