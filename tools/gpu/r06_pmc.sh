#!/bin/bash
# round 6 PMC passes: (1) the GEMM instances of the step -- FETCH / WRITE / MFMA as before plus the L2-side counters this stack has (hit rate, fabric reads by
# destination and size); (2) the attention kernels with the per-dispatch summary (clock <= 2.4 GHz is the sanity check).  Counters in their own passes.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_pmc; mkdir -p $O/profiles
export PMC_MANIFEST=$PWD/$O/pmc_manifest.json
for pm in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $pm | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pm -d $O/pmcg_$tag -o pmc --output-format csv -- python3 tools/pmc_gemm_step.py > $O/pmcg_$tag.log 2>&1; echo "pmcg_$tag rc=$?" >> $O/rc.txt
done
python tools/pmc_to_json.py $O/pmc_manifest.json $O/pmcg_FETCH_SIZE $O/pmcg_WRITE_SIZE $O/pmcg_SQ_VALU_MFMA_BUSY_CYCLES $O/r06_pmc_gemm > $O/pmc_json.log 2>&1; echo "pmc_json rc=$?" >> $O/rc.txt
python tools/pmc_summary.py $O/pmcg_TCC_HIT_sum $O/pmcg_TCC_EA0_RDREQ_sum $O/pmcg_TCC_EA0_RDREQ_128B_sum $O/pmcg_SQ_VALU_MFMA_BUSY_CYCLES --match gemm > $O/r06_pmc_gemm_l2_fabric.csv 2> $O/pmcg_sum.err; echo "pmcg_sum rc=$?" >> $O/rc.txt
for pm in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $pm | cut -d' ' -f1)
  REP=6 timeout 600 rocprofv3 --kernel-trace --pmc $pm -d $O/pmca_$tag -o pmc --output-format csv -- python3 tools/pmc_attn.py > $O/pmca_$tag.log 2>&1; echo "pmca_$tag rc=$?" >> $O/rc.txt
done
python tools/pmc_summary.py $O/pmca_SQ_VALU_MFMA_BUSY_CYCLES $O/pmca_SQ_LDS_BANK_CONFLICT $O/pmca_SQ_WAVE_CYCLES --match attn > $O/r06_pmc_attention.csv 2> $O/pmca_sum.err; echo "pmca_sum rc=$?" >> $O/rc.txt
cp $O/r06_pmc_gemm.csv $O/r06_pmc_gemm.json $O/r06_pmc_gemm_l2_fabric.csv $O/r06_pmc_attention.csv $O/profiles/ 2>/dev/null
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +30M -delete
cat $O/rc.txt; cut -c1-200 $O/r06_pmc_gemm_l2_fabric.csv | head -20; cut -c1-160 $O/r06_pmc_attention.csv | head -20
