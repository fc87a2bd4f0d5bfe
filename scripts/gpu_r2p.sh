set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_lat
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_lat -o lat -- python3 $GRAFT_REPO_ROOT/scripts/quick_timing.py n1024 > $GRAFT_REPO_ROOT/gpurun_out/prof_lat.log 2>&1
cd $GRAFT_REPO_ROOT
grep -v amdgpu.ids gpurun_out/prof_lat.log | grep rep
find gpurun_out/prof_lat -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-220
