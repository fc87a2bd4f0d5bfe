set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "latency_forms or random_small or single_sequence or long_sequence or mid_size" 2>&1 | tail -3
SETS=profile=2,lat_merge=0,lat_split=1 timeout -k 10 300 python scripts/quick_timing.py n1024 n4096 2>&1 | grep -v amdgpu.ids | grep -A1 "rep0"
SETS=profile=2 timeout -k 10 300 python scripts/quick_timing.py n1024 n4096 2>&1 | grep -v amdgpu.ids | grep "rep0"
