#!/bin/bash
# round 4, GPU call N: the grouped 3x3 layers (ResNeXt g32 / super-grouped g8, g16) and the small RTMDet / LeReS layers across the tile families
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04n; mkdir -p $O
CFGS="4 12 6 7 23 26 42 43 46 51 18 38 44" ONLY="4 8 9" timeout 600 python tools/conv_bench8.py > $O/grouped.txt 2>&1
cut -c1-200 $O/grouped.txt | grep -v amdgpu.ids
CFGS="6 7 8 9 11 12 38 39 40 41 42" ONLY="6 5 23 24" timeout 600 python tools/conv_bench8.py > $O/onebyone.txt 2>&1
cut -c1-200 $O/onebyone.txt | grep -v amdgpu.ids
