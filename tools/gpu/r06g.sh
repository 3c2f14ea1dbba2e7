#!/bin/bash
# round 6, call g: does choose_form still pick the best execution form of k_conv_wino4 after the two-square geometry?  (headline shapes, batch 8 / 16)
R=$PWD; O=$R/gpurun_out/r06g; mkdir -p $O
for F in 0 6 3 2; do echo "== CSM_WINO4_FORM=$F (0 = the launcher's own choice)"; CSM_WINO4_FORM=$F timeout 300 python - <<'PY' 2>&1 | grep -E "x[0-9]" | cut -c1-80
import sys; sys.path.insert(0, 'tools')
import wino4_debug as W
for shp in [(8, 40, 40, 256, 256), (8, 40, 40, 512, 256), (16, 23, 23, 512, 512), (16, 23, 23, 512, 256), (16, 23, 23, 256, 128), (16, 45, 45, 256, 512), (16, 45, 45, 512, 128), (16, 45, 45, 256, 256),
            (16, 45, 45, 256, 128), (8, 80, 80, 256, 256), (8, 80, 80, 128, 128), (16, 90, 90, 128, 256), (16, 90, 90, 512, 128), (8, 80, 80, 512, 256), (16, 90, 90, 256, 64), (16, 90, 90, 128, 128)]:
    W.time_layer(*shp, modes=('f4',))
PY
done | tee $O/forms_headline.txt
