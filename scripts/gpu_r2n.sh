set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --workload n4096 --steps 5 --warmup 1 --cpu-budget-s 10 > gpurun_out/r02_bench_n4096_turner.json 2> gpurun_out/n4096.err || { tail -5 gpurun_out/n4096.err; exit 1; }
tail -c 900 gpurun_out/r02_bench_n4096_turner.json; echo
timeout -k 10 300 python bench.py --workload n1024 --model contra --steps 5 --warmup 1 --cpu-budget-s 10 > gpurun_out/r02_bench_n1024_contra.json 2> gpurun_out/n1024.err || { tail -5 gpurun_out/n1024.err; exit 1; }
tail -c 900 gpurun_out/r02_bench_n1024_contra.json; echo
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 1 --warmup 1 --batch-count 600 --no-cpu-baseline --no-n4096 > gpurun_out/r02_bench_torchrun_1rank_batch600.json 2> gpurun_out/torchrun.err || { tail -5 gpurun_out/torchrun.err; exit 1; }
tail -c 300 gpurun_out/r02_bench_torchrun_1rank_batch600.json
