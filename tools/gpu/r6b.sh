#!/bin/bash
# the last commit's tree: attention microbench (all shapes, all generations) and the attention PMC passes
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O/profiles
timeout 900 python tools/bench_attn2.py > $O/attn.log 2> $O/attn.err; rc=$?; echo "attn rc=$rc" >> $O/rc.txt
if [ $rc -eq 0 ] && ! grep -q Traceback $O/attn.log; then grep -v "^/opt/amdgpu" $O/attn.log > $O/profiles/r05_attention_microbench.txt; fi
timeout 900 python tools/bench_attn3_parts.py > $O/parts.log 2>&1; echo "parts rc=$?" >> $O/rc.txt; cat $O/parts.log
ok=0
for pm in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $pm | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pm -d $O/pmca_$tag -o pmc --output-format csv -- python3 tools/pmc_attn.py > $O/pmca_$tag.log 2>&1; rc=$?; echo "pmca_$tag rc=$rc" >> $O/rc.txt; [ $rc -eq 0 ] || ok=1
done
python tools/pmc_summary.py $O/pmca_SQ_VALU_MFMA_BUSY_CYCLES $O/pmca_SQ_LDS_BANK_CONFLICT $O/pmca_SQ_WAVE_CYCLES --match attn > $O/r05_pmc_attention.csv 2> $O/pmca_sum.err; rc=$?; echo "pmca_sum rc=$rc" >> $O/rc.txt
[ $ok -eq 0 ] && [ $rc -eq 0 ] && cp $O/r05_pmc_attention.csv $O/profiles/r05_pmc_attention.csv
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/rc.txt; grep "dkv" $O/r05_pmc_attention.csv | cut -c1-150
