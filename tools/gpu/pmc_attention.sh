#!/bin/bash
# pipe counters of k_attention inside the DPT-BEiT-L core at 672^2 (1765 tokens), one counter set per pass
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r06apmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA")
i=0
for C in "${SETS[@]}"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/s$i -o l -- python $R/tools/zoe_core_profile.py 672 672 > $O/s$i.log 2>&1
done
python - <<PY
import csv,glob,collections
out=open("$O/summary.txt","w")
for d in sorted(glob.glob("$O/*/")):
    agg=collections.defaultdict(lambda:[0,0.0])
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_attention" in r["Kernel_Name"]:
                a=agg[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    line="%s %s" % (d.rstrip("/").split("/")[-1], {k:"%.5g (n=%d)"%(v[1]/v[0],v[0]) for k,v in agg.items()})
    print(line); out.write(line+"\n")
PY
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
