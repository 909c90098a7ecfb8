#!/bin/bash
O=gpurun_out/r5n; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "attention_fwd_bwd and gen2" > $O/pytest_attn.log 2>&1; echo "pytest_attn rc=$?" >> $O/rc.txt
tail -2 $O/pytest_attn.log
timeout 600 python tools/bench_attn3_parts.py > $O/parts.log 2>&1; echo "parts rc=$?" >> $O/rc.txt
cat $O/parts.log
export ATTN=lm REP=3
for pm in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $pm | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pm -d $O/pmca_$tag -o pmc --output-format csv -- python3 tools/pmc_attn.py > $O/pmca_$tag.log 2>&1; echo "pmca_$tag rc=$?" >> $O/rc.txt
done
python tools/pmc_summary.py $O/pmca_SQ_VALU_MFMA_BUSY_CYCLES $O/pmca_SQ_LDS_BANK_CONFLICT $O/pmca_SQ_WAVE_CYCLES $O/pmca_SQ_WAIT_INST_LDS $O/pmca_SQ_INST_CYCLES_VMEM --match attn > $O/pmc_attention_lm.csv 2>$O/pmc_sum.err; echo "sum rc=$?" >> $O/rc.txt
cat $O/pmc_attention_lm.csv
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
