set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "latency_forms or random_small or single_sequence or long_sequence" 2>&1 | tail -4
W=n1024 bash scripts/gpu_r2g.sh latency_mode=1
W=n4096 bash scripts/gpu_r2g.sh latency_mode=1
