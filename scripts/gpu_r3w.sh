#!/bin/bash
# end-of-round checks at HEAD: GPU suite (with the side-stream regression test), soak, the
# driver's exact bench command
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu -s -k "survives_host_entry" 2>&1 | grep -E "tree-order n=2048|passed|failed|rror" | head -5
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1 || { tail -20 gpurun_out/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/pytest_gpu.log
timeout -k 10 600 python tests/soak.py 40 77 > gpurun_out/soak_r03.txt 2>&1; tail -2 gpurun_out/soak_r03.txt
