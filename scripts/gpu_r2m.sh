set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== profile=2"
SETS=profile=2 timeout -k 10 300 python scripts/quick_timing.py n4096 n1024 multi8x2048 2>&1 | grep -v amdgpu.ids | grep rep
