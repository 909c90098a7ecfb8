#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_m; mkdir -p $O
for pm in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  tag=$(echo $pm | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $pm -d $O/p_$tag -o pmc --output-format csv -- python3 tools/pmc_dw.py > $O/p_$tag.log 2>&1; echo "$tag rc=$?" >> $O/rc.txt
done
python tools/pmc_summary.py $O/p_SQ_VALU_MFMA_BUSY_CYCLES $O/p_SQ_WAVE_CYCLES $O/p_SQ_LDS_BANK_CONFLICT $O/p_SQ_WAIT_INST_LDS --match gemm > $O/dw_pmc.csv 2> $O/sum.err
cut -d, -f1,2,3,7,8,11,12,13,14,15 $O/dw_pmc.csv; python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/r06_m/dw_pmc.csv")):
    print(r["kernel"][:50], r["grid"], {k: r[k] for k in r if k.startswith("SQ_")})
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; cat $O/rc.txt
