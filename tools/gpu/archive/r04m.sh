#!/bin/bash
# round 4, GPU call M: fused split-K tail per layer at the single-frame sizes: tuned + forced 64x64 / 128x128 parallel, fused (1) vs reduce kernel (5)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04m; mkdir -p $O
for opt in 5 1; do
  TUNER_OPTIONS=$opt SCALE_N=8 SERIAL=0 CFGS="6 5 44" ONLY="1 5 7 10 14 16 17 19 20 21 22 23 24" timeout 600 python tools/conv_bench8.py > $O/cb1_opt$opt.txt 2>&1
  cut -c1-150 $O/cb1_opt$opt.txt | grep -v amdgpu.ids
done
