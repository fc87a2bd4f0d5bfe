set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_small or two_diagonal or trna or bench_scale or durbin" 2>&1 | tail -3
echo "== top512"
SETS=profile=1 timeout -k 10 300 python scripts/quick_timing.py top512 2>&1 | grep -v amdgpu.ids | grep rep
SETS=profile=1 timeout -k 10 300 python scripts/quick_timing.py top512 2>&1 | grep -v amdgpu.ids | grep rep
