set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "latency_forms or n1024 or trnas or random_small or edge or special" 2>&1 | tee gpurun_out/pytest_gpu_f.log | tail -5
for w in 0 2048 4096 8192; do
W=n1024 bash scripts/gpu_r2g.sh latency_mode=1,lat_inside_waves=$w
W=n4096 bash scripts/gpu_r2g.sh latency_mode=1,lat_inside_waves=$w
done
