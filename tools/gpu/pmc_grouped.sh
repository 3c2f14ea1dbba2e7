#!/bin/bash
# round 6: pipe / memory counters of k_conv_grouped on the three LeReS conv2 shapes at batch 8, one counter set per pass
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r06gpmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp CSM_AUTOTUNE=0
LAYERS=("8 160 160 8 32" "8 80 80 16 32" "8 40 40 32 32")
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS")
for L in "${LAYERS[@]}"; do
  tag=$(echo $L | tr ' ' '_')
  i=0
  for C in "${SETS[@]}"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/${tag}_s$i -o l -- python $R/tools/pmc_grouped.py $L > $O/${tag}_s$i.log 2>&1
  done
done
python - <<PY
import csv,glob,collections
out=open("$O/summary.txt","w")
for d in sorted(glob.glob("$O/*/")):
    agg=collections.defaultdict(lambda:[0,0.0])
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_conv_grouped" in r["Kernel_Name"]:
                a=agg[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    line="%s %s" % (d.rstrip("/").split("/")[-1], {k:"%.5g"%(v[1]/v[0]) for k,v in agg.items()})
    print(line); out.write(line+"\n")
PY
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
