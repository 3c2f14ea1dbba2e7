#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04q; mkdir -p $O
timeout 600 python tools/lanes_probe.py "8x1 4x2 2x4 8x2" > $O/lanes.txt 2>&1
grep -v amdgpu.ids $O/lanes.txt | cut -c1-200
