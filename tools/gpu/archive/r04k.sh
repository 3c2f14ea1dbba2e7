#!/bin/bash
# round 4, GPU call K: timeline of the single-frame step (batch 1): idle gaps, concurrency, per-stream time
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/b1 -- python $R/bench.py --batch 1 --steps 20 --warmup 4 --no-variants --no-cpu-baseline --no-iou --no-roofline > $O/bench_b1.log 2>&1
tail -1 $O/bench_b1.log | cut -c1-300
F=$(find /tmp/b1 -name '*kernel_trace.csv' | head -1)
python $R/tools/trace_gaps.py $F 150 > $O/gaps_b1.txt 2>&1
cat $O/gaps_b1.txt
CSM_OVERLAP_DEPTH=0 timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/b1s -- python $R/bench.py --batch 1 --steps 20 --warmup 4 --no-variants --no-cpu-baseline --no-iou --no-roofline > $O/bench_b1s.log 2>&1
tail -1 $O/bench_b1s.log | cut -c1-200
F=$(find /tmp/b1s -name '*kernel_trace.csv' | head -1)
python $R/tools/trace_gaps.py $F 150 > $O/gaps_b1s.txt 2>&1
head -40 $O/gaps_b1s.txt
