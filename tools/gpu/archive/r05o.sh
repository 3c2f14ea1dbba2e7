#!/bin/bash
# multi-frame warp: the GPU's own timestamps (kernel trace) next to the HIP-event figure of the same calls
OUT=/root/repo/gpurun_out/r05o; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --output-format csv -d $OUT/mf -o mf -- python /root/repo/tools/warp_multiframe_trace.py probe 2>/dev/null | grep "^call" > $OUT/events.txt
cat $OUT/events.txt
python /root/repo/tools/warp_multiframe_trace.py span $OUT/mf | tee $OUT/span.txt
MF_SIZE=2048 MF_K=24 timeout 240 rocprofv3 --kernel-trace --output-format csv -d $OUT/mf2k -o mf -- python /root/repo/tools/warp_multiframe_trace.py probe 2>/dev/null | grep "^call" > $OUT/events_2048.txt
cat $OUT/events_2048.txt
MF_K=24 python /root/repo/tools/warp_multiframe_trace.py span $OUT/mf2k | tee $OUT/span_2048.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
