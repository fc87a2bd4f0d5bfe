set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "latency_forms" 2>&1 | tail -4
for v in 0 2; do
W=n1024 bash scripts/gpu_r2g.sh latency_mode=1,lat_inside=$v
W=n4096 bash scripts/gpu_r2g.sh latency_mode=1,lat_inside=$v
done
