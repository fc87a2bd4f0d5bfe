set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_n4096
cd $R
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_n4096 -o run -- python3 scripts/quick_timing.py n4096 > gpurun_out/prof_n4096.log 2>&1 || tail -5 gpurun_out/prof_n4096.log
find gpurun_out/prof_n4096 -name "*kernel_stats*" | head -3
F=$(find gpurun_out/prof_n4096 -name "*kernel_stats.csv" | head -1)
cut -c1-200 $F | head -20
