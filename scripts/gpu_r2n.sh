set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in multi16x1024 multi8x2048 multi4x1024 multi32x512; do
for v in 1024 2048 3072 4096; do
W=$w bash scripts/gpu_r2g.sh latency_mode=1,lat_inside=2,lat_e_waves=$v
done
done
