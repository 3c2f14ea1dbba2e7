#!/bin/bash
# rocprofv3 per-kernel statistics of the DPT-BEiT-L core at the reference's ZoeDepth size (672 x 672, 1765 tokens)
OUT=/root/repo/gpurun_out/r05q; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 220 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/zoe -o zoe -- python /root/repo/tools/zoe_core_profile.py 672 672 > $OUT/zoe.log 2>&1
grep -v "amdgpu.ids" $OUT/zoe.log | tail -3
f=$(find $OUT/zoe -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
