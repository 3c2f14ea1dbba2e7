#!/bin/bash
# round 4, GPU call Y: attention with the second block of every CU started late (s_sleep 32 / 64 x 64 cycles): are co-resident blocks in phase?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04y; mkdir -p $O
for lib in "" tools/scratch_libs/libcsm355_dephase32.so tools/scratch_libs/libcsm355_dephase64.so; do
  for n in 1 2; do
    echo "lib=$lib n=$n" | tee -a $O/dephase.txt
    CSM_LIB=${lib:+$R/$lib} timeout 300 python tools/zoe_core_profile.py 672 672 $n 2>&1 | grep attention | tail -1 | tee -a $O/dephase.txt
  done
done
