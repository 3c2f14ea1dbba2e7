#!/bin/bash
# round 4, GPU call O: 40 x 40 / 20 x 20 / 23 x 23 / 45 x 45 feature maps: 8-wide against 16-wide patch tiles (tile quantisation of the small maps)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04o; mkdir -p $O
for ser in 1 0; do
SERIAL=$ser CFGS="24 25 26 18 19 21 27 44 45 47 6 7" ONLY="10 16 17 22" timeout 600 python tools/conv_bench8.py > $O/small_maps_ser$ser.txt 2>&1
cut -c1-220 $O/small_maps_ser$ser.txt | grep -v amdgpu.ids
done
