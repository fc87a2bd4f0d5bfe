#!/bin/bash
# which kernels carry the tree-order mode on a batch (128 longest sequences of the 10k batch)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_tree_batch
rm -rf $OUT && mkdir -p $OUT
SETS=summation_mode=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/scripts/quick_timing.py top128 > $OUT/traced.txt 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
find $OUT/trace -name "*kernel_trace.csv" -delete
grep rep0 $OUT/traced.txt
head -14 $OUT/kernel_stats.csv | cut -c1-260
