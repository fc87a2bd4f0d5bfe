set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in multi64x256 multi16x256 multi6x76; do
W=$w bash scripts/gpu_r2g.sh latency_mode=1
done
