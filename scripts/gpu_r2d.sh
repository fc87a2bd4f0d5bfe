set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
W=${1:-top512}
shift || true
for cfg in "$@"; do
  echo "== full sweep, $cfg"
  SETS=$cfg timeout -k 10 200 python scripts/quick_timing.py $W 2>&1 | grep -v amdgpu.ids
done
