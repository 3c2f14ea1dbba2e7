#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04l; mkdir -p $O
python tools/scratch/dbg_fused.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "fused_splitk" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for opt in 5 1; do
CSM_TUNER_OPTIONS=$opt CSM_OVERLAP_DEPTH=0 timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/b1s$opt -- python $R/bench.py --batch 1 --steps 20 --warmup 4 --no-variants --no-cpu-baseline --no-iou --no-roofline > $O/bench_b1s_$opt.log 2>&1
tail -1 $O/bench_b1s_$opt.log | cut -c1-200
F=$(find /tmp/b1s$opt -name '*kernel_trace.csv' | head -1)
python $R/tools/trace_gaps.py $F 150 > $O/gaps_b1s_$opt.txt 2>&1
grep -A 14 "kernels by time" $O/gaps_b1s_$opt.txt; head -3 $O/gaps_b1s_$opt.txt
done
