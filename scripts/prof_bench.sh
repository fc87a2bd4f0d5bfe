# rocprofv3 summaries of the bench command for profiles/ (run on the GPU box via gpurun).
# usage: bash scripts/prof_bench.sh <tag> [bench args]
set -e
TAG=${1:-r01}; shift || true
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_bench_$TAG
rm -rf $OUT && mkdir -p $OUT
# 1) plain bench (the reported line)
python3 $R/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
tail -c 1500 $OUT/bench.json
# 2) kernel trace + stats of the same command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py "$@" --no-cpu-baseline > $OUT/bench_traced.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
find $OUT/trace -name "*kernel_trace.csv" -delete
cat $OUT/kernel_stats.csv
