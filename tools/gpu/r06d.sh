#!/bin/bash
# round 6, call d: the row-split execution forms of k_conv_wino4: bit-exactness per form, then layer times per form (one process per form)
R=$PWD; O=$R/gpurun_out/r06d; mkdir -p $O
for F in ${FORMS:-1 2 3 6}; do CSM_WINO4_FORM=$F timeout 300 python tools/wino4_debug.py check 2>&1 | grep -E "ALL|SOME|FAIL" | sed "s/^/form $F: /"; done
for F in ${FORMS:-6 3 2 1}; do echo "== CSM_WINO4_FORM=$F"; CSM_WINO4_FORM=$F timeout 400 python tools/wino4_debug.py forms 2>&1 | grep -E "x[0-9]" | cut -c1-150; done | tee $O/forms_${TAG:-a}.txt
