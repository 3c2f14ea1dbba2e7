#!/bin/bash
# round 4, GPU call V: counters of the ResNeXt grouped 3x3 (32 groups of 32 channels at 40 x 40 x 8) under the plain DMA tile (12) and the patch tile (43)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04v; mkdir -p $O
cd /tmp && export TMPDIR=/tmp CSM_AUTOTUNE=0
LAYERS=("8 40 40 1024 1024 3 12 32" "8 40 40 1024 1024 3 43 32")
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum")
for L in "${LAYERS[@]}"; do
  tag=$(echo $L | tr ' ' '_')
  for C in "${SETS[@]}"; do
    c1=$(echo $C | cut -d' ' -f1)
    timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/${tag}_$c1 -o l -- python $R/tools/pmc_layer.py $L > $O/${tag}_$c1.log 2>&1
  done
done
python - <<PY
import csv,glob,collections
out=open("$O/summary.txt","w")
for tag in ("8_40_40_1024_1024_3_12_32","8_40_40_1024_1024_3_43_32"):
    agg=collections.defaultdict(lambda:[0,0.0]); dur=[]
    for f in glob.glob("$O/"+tag+"_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_conv" in r["Kernel_Name"]:
                a=agg[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    for f in glob.glob("$O/"+tag+"_*/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_conv" in r["Kernel_Name"]: dur.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    line="%s us=%.1f %s" % (tag, sum(dur)/max(len(dur),1)/1e3, {k:"%.4g"%(v[1]/v[0]) for k,v in sorted(agg.items())})
    print(line); out.write(line+"\n")
PY
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
