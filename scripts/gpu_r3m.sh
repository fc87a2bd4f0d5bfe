#!/bin/bash
# round 3, third session, call 1: state check + ubench of finite-operand lse forms + A/B of the
# finite-operand fold chunks (librnamc_base.so = HEAD before them)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
timeout -k 10 300 python scripts/quick_timing.py top512 > gpurun_out/qt_top512_new.txt 2>&1 || { tail -5 gpurun_out/qt_top512_new.txt; exit 1; }
cat gpurun_out/qt_top512_new.txt
RNAMC_LIB=$PWD/rna_algos_amd/librnamc_base.so timeout -k 10 300 python scripts/quick_timing.py top512 > gpurun_out/qt_top512_base.txt 2>&1 || exit 1
cat gpurun_out/qt_top512_base.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -5 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 420 ./scripts/ubench/valu_rate > gpurun_out/ubench_valu_rate.txt 2>&1 || exit 1
timeout -k 10 300 ./scripts/ubench/lse_fast > gpurun_out/ubench_lse_fast.txt 2>&1 || exit 1
timeout -k 10 300 ./scripts/ubench/lse_fast_noslp > gpurun_out/ubench_lse_fast_noslp.txt 2>&1 || exit 1
cat gpurun_out/ubench_lse_fast_noslp.txt
timeout -k 10 300 python bench.py --workload n4096 --steps 5 --warmup 1 > gpurun_out/bench_n4096_turner.json 2> gpurun_out/bench_n4096_turner.err || exit 1
timeout -k 10 300 python bench.py --workload n4096 --model contra --steps 5 --warmup 1 > gpurun_out/bench_n4096_contra.json 2> gpurun_out/bench_n4096_contra.err || exit 1
