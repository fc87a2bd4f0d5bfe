#!/bin/bash
# why is the n = 4096 tree leg of the BATCH bench 3x slower than in --workload n4096?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/tree_leg_probe.txt; rm -f $L
run() {
  echo "== $*" >> $L
  timeout -k 10 200 python bench.py --batch-count 600 --steps 1 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  tree', d.get('n4096_tree', {}).get('ms_all_calls'), 'ref', d.get('ms_per_seq_n4096_calls'), 'value', round(d['value']))" >> $L
}
run
run --no-transfers
run --no-kernel-timing
run --no-transfers --no-kernel-timing
run --group-ws-gb 8
cat $L
