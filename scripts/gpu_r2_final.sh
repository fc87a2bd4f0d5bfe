# end-of-round check: full GPU suite, smoke, soak
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( while true; do sleep 60; echo "[alive $(date +%T)]"; done ) &
KEEP=$!
trap "kill $KEEP 2>/dev/null" EXIT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 900 python tests/soak.py 40 777 > gpurun_out/soak_r02.log 2>&1 || { tail -5 gpurun_out/soak_r02.log; exit 1; }
tail -2 gpurun_out/soak_r02.log
