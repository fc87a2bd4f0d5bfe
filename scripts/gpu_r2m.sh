set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lds_staged" 2>&1 | tail -4
bash scripts/gpu_r2d.sh top512 head_lds=0 head_lds=2
