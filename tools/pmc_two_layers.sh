#!/bin/bash
# PMC comparison of a short-K 1x1 layer and a long-K 3x3 layer under the 128x128 LDS-DMA tile (cfg 8): where do the cycles go?
cd /tmp && export TMPDIR=/tmp CSM_AUTOTUNE=0
OUT=/root/repo/gpurun_out/pmc2; mkdir -p $OUT
for L in "8 48 32 1024 1024 1" "8 160 160 256 256 3"; do
  tag=$(echo $L | tr ' ' '_')
  for C in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
    c1=$(echo $C | cut -d' ' -f1)
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${tag}_$c1 -o l -- python /root/repo/tools/pmc_layer.py $L 8 > /dev/null 2>&1
  done
done
python - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$OUT/*")):
    agg=collections.defaultdict(lambda:[0,0.0])
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_conv" in r["Kernel_Name"]:
                a=agg[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    print(d.split("/")[-1], {k:"%.4g"%(v[1]/v[0]) for k,v in agg.items()})
PY
