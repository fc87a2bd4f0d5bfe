cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_contra
rm -rf $OUT && mkdir -p $OUT
GSIZES=1024 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/scripts/quick_timing.py batch1000 > $OUT/run.log 2> $OUT/err.log
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
find $OUT/trace -name "*kernel_trace.csv" -delete
grep rnamc $OUT/kernel_stats.csv | cut -c1-200 | head -14
