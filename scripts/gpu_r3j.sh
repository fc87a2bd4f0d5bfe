# round 3: upper bound of what removing the 2-loop blocks' HBM traffic could gain (verdict item 4):
# the same workloads through the product library and through a build whose probes all re-read the
# cell's first probe (librnamc_resident.so, -DRNAMC_PROBE_RESIDENT; results wrong, timing only)
L=gpurun_out/r03_probe_resident.log
rm -f $L
for lib in librnamc.so librnamc_resident.so; do
  echo "=== $lib" >> $L
  RNAMC_LIB=$PWD/rna_algos_amd/$lib timeout -k 10 500 python scripts/quick_timing.py top512 bot2000 2>&1 | grep -v amdgpu.ids >> $L
  CONTRA=0 GSIZES=1024 RNAMC_LIB=$PWD/rna_algos_amd/$lib timeout -k 10 300 python scripts/quick_timing.py batch1000 2>&1 | grep -v amdgpu.ids >> $L
done
cat $L
