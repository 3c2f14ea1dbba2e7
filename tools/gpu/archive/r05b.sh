#!/bin/bash
# round 5, call b: where do k_conv_wino's cycles go?  ablation / option instantiations (dev library) + SQ / LDS counters of the shipped kernel
R=$PWD; O=$R/gpurun_out/r05b; mkdir -p $O
export CSM_SYNTHETIC_WEIGHTS=1
CSM_LIB=$R/cartoonsegmentation_amd/libcsm355_dev.so timeout 600 python tools/wino_debug.py variants > $O/variants.txt 2>&1; cat $O/variants.txt | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp CSM_AUTOTUNE=0 CSM_WINO_MIN_PIXELS=0
for L in "8 160 160 256 256 3" "16 360 360 64 64 3"; do
  tag=$(echo $L | tr ' ' '_')
  for C in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
    c1=$(echo $C | cut -d' ' -f1)
    timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_${tag}_$c1 -o l -- python $R/tools/pmc_layer.py $L > $O/pmc_${tag}_$c1.log 2>&1
  done
done
python - <<PY > $O/pmc_summary.txt
import csv,glob,collections
for d in sorted(glob.glob("$O/pmc_*")):
    if d.endswith('.log'): continue
    agg=collections.defaultdict(lambda:[0,0.0])
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_conv_wino" in r["Kernel_Name"]:
                a=agg[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    print(d.split("/")[-1], {k:"%.5g"%(v[1]/v[0]) for k,v in agg.items()})
PY
cat $O/pmc_summary.txt
find $O -name "*.csv" -size +200k -delete
cd $R
# (zoe pipeline test: next call)
timeout 300 python -m pytest tests/test_gpu_warp.py -q -k "pointwise" > $O/pytest_warp.txt 2>&1; tail -5 $O/pytest_warp.txt
