#!/bin/bash
# round 4, GPU call P: timeline of the default 8-frame step: idle gaps, concurrency
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/b8 -- python $R/bench.py --steps 8 --warmup 2 --no-variants --no-cpu-baseline --no-iou --no-roofline > $O/bench_b8.log 2>&1
grep '"metric"' $O/bench_b8.log | cut -c1-200
F=$(find /tmp/b8 -name '*kernel_trace.csv' | head -1)
python $R/tools/trace_gaps.py $F 500 > $O/gaps_b8.txt 2>&1
cat $O/gaps_b8.txt
