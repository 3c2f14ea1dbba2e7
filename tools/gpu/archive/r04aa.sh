#!/bin/bash
# round 4, GPU call AA: the default bench line at HEAD (16 steps, 4 warm-up)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04aa; mkdir -p $O
T0=$(date +%s); timeout 900 python bench.py 2>/dev/null | tail -1 > $O/bench_frame.json; head -c 300 $O/bench_frame.json; echo
echo "python bench.py (default flags, all variants): $(( $(date +%s) - T0 )) s wall" | tee $O/bench_seconds.txt
