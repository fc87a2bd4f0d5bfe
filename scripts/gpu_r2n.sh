set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CONTRA=0 GSIZES=64,128,256,512,1024,8192 SETS=profile=1 timeout -k 10 600 python scripts/quick_timing.py batch2000 2>&1 | grep -v amdgpu.ids | grep rep
