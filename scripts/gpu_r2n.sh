set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in multi128x256 multi256x256 multi64x512 multi32x1024; do
for v in 0 2; do
W=$w bash scripts/gpu_r2g.sh latency_mode=$v
done
done
