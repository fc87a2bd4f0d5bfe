# One GPU call that refreshes everything judged: tests, torchrun path, traffic counters.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
# the N>1 launcher path with one rank (nccl init, barrier, all_reduce of the step time)
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 1 --warmup 0 --batch-count 300 --no-cpu-baseline > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err || { tail -5 gpurun_out/bench_torchrun1.err; exit 1; }
tail -c 400 gpurun_out/bench_torchrun1.json
bash scripts/prof_traffic.sh r01 --batch-count 1000 > gpurun_out/prof_traffic_r01.log 2>&1 || tail -5 gpurun_out/prof_traffic_r01.log
timeout -k 10 300 python bench.py --workload n4096 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_n4096.json 2>/dev/null; tail -c 600 gpurun_out/bench_n4096.json
timeout -k 10 300 python bench.py --workload n1024 --model contra --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_n1024_contra.json 2>/dev/null; tail -c 300 gpurun_out/bench_n1024_contra.json
