#!/bin/bash
# round 4, GPU call Z: DMA kernels with magic-number row set-up + row-pointer epilogue: bit-exactness of every tile, then the bench's conv layers
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04z; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_nets.py tests/test_gpu_fullsize.py -x -q -m gpu -k "tile_configurations or repeated_runs or bit_exact or conv" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 900 python tools/conv_bench8.py > $O/cb8.txt 2>&1
cut -c1-120 $O/cb8.txt | grep -v amdgpu.ids | tail -40
