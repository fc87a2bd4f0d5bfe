# rocprofv3 kernel stats of the tree-order mode on the 10k batch (run on the GPU box via gpurun).
# usage: bash scripts/prof_tree_batch.sh <tag> [batch-count] [model]
set -e
TAG=${1:-r04}; CNT=${2:-10000}; MODEL=${3:-turner}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_tree_batch_$TAG
rm -rf $OUT && mkdir -p $OUT
ARGS="--summation tree --batch-count $CNT --model $MODEL --steps 1 --warmup 1 --no-cpu-baseline --no-n4096 --no-transfers"
python3 $R/bench.py $ARGS > $OUT/plain.json 2> $OUT/plain.err || { tail -5 $OUT/plain.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/traced.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
find $OUT/trace -name "*kernel_trace.csv" -delete
python3 -c "
import json; j = json.load(open('$OUT/plain.json')); print(j['value'], j['ms_per_step'], j['roofline'].get('ms_inside_per_step'), j['roofline'].get('ms_outside_per_step'))"
head -14 $OUT/kernel_stats.csv
