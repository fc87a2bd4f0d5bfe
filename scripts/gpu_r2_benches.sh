# end-of-round records: the driver's command, the two single-sequence configs, the launcher path
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( while true; do sleep 60; echo "[alive $(date +%T)]"; done ) &
KEEP=$!
trap "kill $KEEP 2>/dev/null" EXIT
python3 bench.py --workload n4096 --steps 5 --warmup 1 > gpurun_out/r02_bench_n4096_turner.json 2> gpurun_out/n4096.err || { tail -5 gpurun_out/n4096.err; exit 1; }
tail -c 400 gpurun_out/r02_bench_n4096_turner.json; echo
python3 bench.py --workload n1024 --model contra --steps 5 --warmup 1 > gpurun_out/r02_bench_n1024_contra.json 2> gpurun_out/n1024.err || { tail -5 gpurun_out/n1024.err; exit 1; }
tail -c 400 gpurun_out/r02_bench_n1024_contra.json; echo
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --batch-count 600 --steps 1 --warmup 1 > gpurun_out/r02_bench_torchrun_1rank_batch600.json 2> gpurun_out/torchrun.err || { tail -5 gpurun_out/torchrun.err; exit 1; }
tail -c 300 gpurun_out/r02_bench_torchrun_1rank_batch600.json; echo
SECONDS=0
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_driver_command.json 2> gpurun_out/driver.err || { tail -5 gpurun_out/driver.err; exit 1; }
echo "driver command wall: $SECONDS s"
tail -c 300 gpurun_out/r02_bench_driver_command.json; echo
