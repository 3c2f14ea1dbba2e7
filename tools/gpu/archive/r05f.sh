#!/bin/bash
# round 5, call f (wino8 default, residual fix): the Winograd rule (>= 80 x 80 output pixels per sample) in the nets: layer profile at batch 8, the bench line, the GPU suite
O=gpurun_out/r05f; mkdir -p $O
export CSM_SYNTHETIC_WEIGHTS=1
LP_BATCH=8 timeout 900 python tools/layer_profile.py > $O/lp_b8.txt 2>&1; grep -v amdgpu.ids $O/lp_b8.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, {k: d["roofline"][k] for k in ("achieved", "frac", "direct_equivalent_tflops", "conv_ms_per_step", "winograd_ms_per_step", "winograd_launches_per_step")})
print(d["roofline"]["per_net"]); print(d.get("mask_iou_vs_oracle")); print(d.get("batch1"))
PY
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt
