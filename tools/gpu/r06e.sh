#!/bin/bash
# round 6, call e: row-split forms + small-map rule: tests, then the bench with / without the small-map part of the rule
R=$PWD; O=$R/gpurun_out/r06e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_winograd4.py -x -q 2>&1 | tail -3
for S in 500 6400; do
  CSM_WINO4_SMALL_MIN_PIXELS=$S timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_small$S.json
  python - <<PY
import json
d=json.load(open("$O/bench_small$S.json"))
r=d["roofline"]
print("small_min_pixels=$S: value", d["value"], "ms/step", d["ms_per_step"], "conv ms", r["conv_ms_per_step"], "wino ms", r["winograd_ms_per_step"], "batch1", d["batch1"]["frames_per_s"], "b1 4 in flight", d["batch1"].get("frames_per_s_4_in_flight"), "iou", d["mask_iou_vs_oracle"]["iou_min"])
for k,v in r["per_class"].items(): print("   ",k,v["launches_per_step"],v["ms_per_step"],v["frac_of_mfma_peak"])
PY
done
