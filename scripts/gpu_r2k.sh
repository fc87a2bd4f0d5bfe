set -e
cd $GRAFT_REPO_ROOT
for v in f1 f2; do
  echo "== lib $v"
  RNAMC_LIB=$PWD/rna_algos_amd/librnamc_$v.so SETS=latency_mode=1,profile=2 timeout -k 10 200 python scripts/quick_timing.py n1024 2>&1 | grep -v amdgpu.ids | grep -A1 "contra=False rep1"
done
