# k_head timing: whole sweeps and the head role alone, gather form against LDS-staged form
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -s -C rna_algos_amd/csrc DEBUG_KNOBS=1 OUT=../librnamc_dbg.so 2>&1 | grep -E "error" || true
export RNAMC_LIB=$PWD/rna_algos_amd/librnamc_dbg.so
W=${1:-top512}
for cfg in "head_lds=0" "head_lds=1" "head_lds=1,head_wmax_in=320,head_wmax_out=320"; do
  echo "== full sweep, $cfg"
  SETS=$cfg timeout -k 10 200 python scripts/quick_timing.py $W 2>&1 | grep -v amdgpu.ids
done
for cfg in "head_lds=0" "head_lds=1"; do
  echo "== outside: 2-loop half alone (ROLES=43), $cfg"
  ROLES=43 SETS=$cfg timeout -k 10 200 python scripts/quick_timing.py $W 2>&1 | grep -v amdgpu.ids
  echo "== inside: pair blocks alone (ROLES=2), $cfg"
  ROLES=2 SETS=$cfg timeout -k 10 200 python scripts/quick_timing.py $W 2>&1 | grep -v amdgpu.ids
done
