#!/bin/bash
# round-3 records of the Turner batch at the round's final kernels: bench line, rocprofv3 kernel
# stats of the same command, PMC traffic of the 1000-sequence profiling batch
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 700 bash scripts/prof_bench.sh r03 --steps 3 --warmup 2 > gpurun_out/prof_bench_r03.log 2>&1 || { tail -20 gpurun_out/prof_bench_r03.log; exit 1; }
tail -12 gpurun_out/prof_bench_r03.log | cut -c1-220
timeout -k 10 400 bash scripts/prof_traffic.sh r03 --batch-count 1000 > gpurun_out/prof_traffic_r03.log 2>&1 || { tail -20 gpurun_out/prof_traffic_r03.log; exit 1; }
tail -5 gpurun_out/prof_traffic_r03.log | cut -c1-300
