#!/bin/bash
# round 5, call i: rule threshold 45 x 45 (ISNet's 45^2 layers on the 16 x 16-pixel geometry): layer profile, batch-1 profile, bench without variants
O=gpurun_out/r05i; mkdir -p $O
export CSM_SYNTHETIC_WEIGHTS=1
LP_BATCH=8 timeout 900 python tools/layer_profile.py isnet > $O/lp_b8_isnet.txt 2>&1; grep -v amdgpu.ids $O/lp_b8_isnet.txt
LP_BATCH=1 timeout 900 python tools/layer_profile.py isnet > $O/lp_b1_isnet.txt 2>&1; grep -v amdgpu.ids $O/lp_b1_isnet.txt | head -12
CSM_WINO_MIN_PIXELS=6400 LP_BATCH=1 timeout 900 python tools/layer_profile.py isnet > $O/lp_b1_isnet_6400.txt 2>&1; grep -v amdgpu.ids $O/lp_b1_isnet_6400.txt | head -12
timeout 900 python bench.py --no-variants --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, {k: d["roofline"][k] for k in ("achieved", "frac", "direct_equivalent_tflops", "conv_ms_per_step", "winograd_ms_per_step", "winograd_launches_per_step")}, d.get("mask_iou_vs_oracle", {}).get("iou_min"))
PY
