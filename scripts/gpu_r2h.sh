set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -s -C rna_algos_amd/csrc DEBUG_KNOBS=1 OUT=../librnamc_dbg.so 2>&1 | grep -E "error" || true
export RNAMC_LIB=$PWD/rna_algos_amd/librnamc_dbg.so
for r in 15 11 27 59; do
  for cfg in "$@"; do
    echo "== ROLES=$r $cfg"
    ROLES=$r SETS=$cfg timeout -k 10 200 python scripts/quick_timing.py n4096 2>&1 | grep -v amdgpu.ids | grep "contra=False rep1"
  done
done
