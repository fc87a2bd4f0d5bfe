set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "trnas or random_small or latency_forms or n1024 or n4096" 2>&1 | tail -4
W=n1024 bash scripts/gpu_r2g.sh latency_mode=1
W=n4096 bash scripts/gpu_r2g.sh latency_mode=1
