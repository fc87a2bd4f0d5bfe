# round 2 evidence: the bench line, rocprofv3 kernel stats of the same command, PMC traffic
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
( while true; do sleep 60; echo "[alive $(date +%T)]"; done ) &
KEEP=$!
trap "kill $KEEP 2>/dev/null" EXIT
cd $R
# 1) the reported line, at a K/W that fits this call
python3 bench.py --steps 3 --warmup 2 > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err || { tail -5 gpurun_out/r02_bench.err; exit 1; }
tail -c 600 gpurun_out/r02_bench.json; echo
# 2) kernel trace + stats of the bench command (one timed pass, the kernel-timing warm-up pass)
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_bench_r02
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-transfers > $OUT/bench_traced.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r02_bench_kernel_stats.csv
find $OUT/trace -name "*kernel_trace.csv" -delete
cut -c1-160 $R/gpurun_out/r02_bench_kernel_stats.csv | head -24
cp $OUT/bench_traced.json $R/gpurun_out/r02_bench_traced.json
# 3) HBM traffic counters (separate --pmc passes), 1000-sequence batch
cd $R
bash scripts/prof_traffic.sh r02 --batch-count 1000 > gpurun_out/prof_traffic_r02.log 2>&1 || { tail -5 gpurun_out/prof_traffic_r02.log; exit 1; }
cp gpurun_out/prof_traffic_r02/traffic.json gpurun_out/r02_traffic_batch1000.json
tail -5 gpurun_out/prof_traffic_r02.log
