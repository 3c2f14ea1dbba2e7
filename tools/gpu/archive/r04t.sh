#!/bin/bash
# round 4, GPU call T: pipe-utilisation counters of k_attention (DPT-BEiT-L core at 672^2, 1765 tokens), one counter set per pass
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC")
for C in "${SETS[@]}"; do
  c1=$(echo $C | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/att_$c1 -o l -- python $R/tools/zoe_core_profile.py 672 672 > $O/att_$c1.log 2>&1
done
python - <<PY
import csv,glob,collections
out=open("$O/summary.txt","w")
agg=collections.defaultdict(lambda:[0,0.0])
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_attention" in r["Kernel_Name"]:
            a=agg[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
for k,v in sorted(agg.items()):
    line="%-28s %.5g  (x%d)" % (k, v[1]/v[0], v[0]); print(line); out.write(line+"\n")
PY
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
