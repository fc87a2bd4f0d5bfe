set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "$@"; do
  echo "== $cfg"
  SETS=$cfg timeout -k 10 200 python scripts/quick_timing.py ${W:-n4096} 2>&1 | grep -v amdgpu.ids | grep -A1 "rep1"
done
