set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
W=${1:-top512}
for v in "" _nt1 _nt2 _nt4 _nt7; do
  echo "== lib$v"
  RNAMC_LIB=$PWD/rna_algos_amd/librnamc$v.so timeout -k 10 200 python scripts/quick_timing.py $W 2>&1 | grep -v amdgpu.ids
done
