set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in multi8x2048 multi16x1024; do
for v in 2048 3072 4096 6144; do
W=$w bash scripts/gpu_r2g.sh lat_e_waves=$v
done
done
