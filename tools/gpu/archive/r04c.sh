#!/bin/bash
# round 4, GPU call C: patch-kernel swizzle without bank conflicts, N-grouped tile order (on / off), tuner with warm-up + interleaved finalists
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04c; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "tile_configurations or repeated_runs or conv_bit_exact" > $O/pytest_nets.log 2>&1
tail -3 $O/pytest_nets.log
CF="6 7 9 11 38 39 41 44 45 47 48"
CFGS="$CF" timeout 900 python tools/conv_bench8.py > $O/cb8_group.txt 2>&1
TUNER_OPTIONS=3 CFGS="$CF" timeout 900 python tools/conv_bench8.py > $O/cb8_nogroup.txt 2>&1
tail -1 $O/cb8_group.txt $O/cb8_nogroup.txt
LP_BATCH=8 timeout 900 python tools/layer_profile.py > $O/lp8.txt 2>&1
grep "^==" $O/lp8.txt
cd /tmp && export TMPDIR=/tmp CSM_AUTOTUNE=0
for spec in "8 160 160 256 256 3 45|SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "8 40 40 1024 1024 1 6|FETCH_SIZE" "8 40 40 1024 1024 1 6|TCC_HIT_sum TCC_MISS_sum" "8 40 40 1024 1024 1 6|GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
  L=${spec%%|*}; C=${spec##*|}; tag=$(echo $L | tr ' ' '_')_$(echo $C | cut -d' ' -f1)
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$tag -o l -- python $R/tools/pmc_layer.py $L > $O/$tag.log 2>&1
done
python - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$O/*/")):
    agg=collections.defaultdict(lambda:[0,0.0])
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_conv" in r["Kernel_Name"]:
                a=agg[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    print(d.rstrip("/").split("/")[-1], {k:"%.4g"%(v[1]/v[0]) for k,v in agg.items()})
PY
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
