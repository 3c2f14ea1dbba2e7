#!/bin/bash
# round 4, GPU call L: split-K tail fused into the conv kernels (last-arriving block) -- parity / stress, then the single-frame step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "fused_splitk or stress or tile_configurations or bit_exact" > $O/pytest_nets.log 2>&1
tail -5 $O/pytest_nets.log
for opt in 5 1; do
  CSM_TUNER_OPTIONS=$opt timeout 400 python bench.py --batch 1 --steps 30 --warmup 5 --no-variants --no-cpu-baseline --no-iou --no-roofline > $O/bench_b1_opt$opt.log 2>&1
  tail -1 $O/bench_b1_opt$opt.log | cut -c1-160
done
for opt in 5 1; do
  CSM_TUNER_OPTIONS=$opt timeout 400 python bench.py --steps 6 --warmup 2 --no-variants --no-cpu-baseline --no-iou --no-roofline > $O/bench_b8_opt$opt.log 2>&1
  tail -1 $O/bench_b8_opt$opt.log | cut -c1-160
done
