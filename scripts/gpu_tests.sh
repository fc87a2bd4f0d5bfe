set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee gpurun_out/smoke.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tee gpurun_out/pytest_gpu.log
