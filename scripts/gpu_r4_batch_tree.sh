# Round-4 records of the tree-order mode's BATCH form (rnamc_tree_lane.h, rnamc_tree_mx.h), one gpurun call:
#   bash scripts/gpu_r4_batch_tree.sh        (writes under gpurun_out/; copy what is to be judged into profiles/)
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 bash scripts/prof_traffic_tree_batch.sh r04 0 > gpurun_out/traffic_tree_batch_r04.log 2>&1
tail -3 gpurun_out/traffic_tree_batch_r04/plain.txt
timeout -k 10 200 python bench.py --summation tree --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_batch_tree_final.json 2> gpurun_out/r04_bench_batch_tree_final.err
timeout -k 10 200 python bench.py --summation tree --model contra --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_batch_tree_contra.json 2> gpurun_out/r04_bench_batch_tree_contra.err
timeout -k 10 400 python bench.py --steps 2 --warmup 1 > gpurun_out/r04_bench_short_default.json 2> gpurun_out/r04_bench_short_default.err
timeout -k 10 200 python scripts/tree_ms.py 1024 2048 4096 8192 16384 > gpurun_out/r04_tree_ms_mx.txt 2>&1
timeout -k 10 200 python scripts/lane_check.py > gpurun_out/lane_check.txt 2>&1; tail -1 gpurun_out/lane_check.txt
timeout -k 10 200 python scripts/mx_check.py > gpurun_out/mx_check.txt 2>&1; tail -1 gpurun_out/mx_check.txt
timeout -k 10 100 python scripts/gen_batch_check.py > gpurun_out/gen_batch_check.txt 2>&1; tail -3 gpurun_out/gen_batch_check.txt
