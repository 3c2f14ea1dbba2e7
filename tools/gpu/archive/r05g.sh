#!/bin/bash
# round 5, call g: multi-frame warp call, poisoned-state detection, bounded core cache, batched zoe; the bench line with its variants
O=gpurun_out/r05g; mkdir -p $O
export CSM_SYNTHETIC_WEIGHTS=1
timeout 900 python -m pytest tests/test_gpu_warp.py tests/test_gpu_dpt_beit.py tests/test_gpu_nets.py tests/test_gpu_kenburns.py -m gpu -q -x -k "multi_frame or bounded or percentiles or zoe or pointwise" > $O/pytest_sel.txt 2>&1; tail -12 $O/pytest_sel.txt
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, {k: d["roofline"][k] for k in ("achieved", "frac", "direct_equivalent_tflops", "conv_ms_per_step", "winograd_ms_per_step")})
v = d.get("variants", {})
print("batch1", v.get("batch1")); print("batch16", v.get("batch16"))
print("zoe", {k: v.get("zoe_depth_batch1", {}).get(k) for k in ("frames_per_s", "ms_per_frame", "batch8")})
for k, x in v.get("zoe_depth_batch1", {}).items():
    if k.startswith("core_"): print(k, x)
for k, x in v.get("warp_chain", {}).items(): print(k, x)
print("video", v.get("video")); print("cpu", d.get("cpu_baseline"))
PY
