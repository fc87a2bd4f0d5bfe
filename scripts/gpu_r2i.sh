set -e
cd $GRAFT_REPO_ROOT
export RNAMC_LIB=$PWD/rna_algos_amd/librnamc_dbg.so
SETS=latency_mode=1 timeout -k 10 200 python scripts/quick_timing.py n4096 n1024 2>&1 | grep -v amdgpu.ids
