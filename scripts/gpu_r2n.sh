set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in multi128x256 multi64x1024 multi512x256 multi2048x256; do
for v in 0 2; do
W=$w bash scripts/gpu_r2g.sh latency_mode=$v
done
done
