cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2a; mkdir -p $O
nproc > $O/host.txt; free -g >> $O/host.txt
timeout 1500 python -m pytest tests/test_widths_gpu.py -x -q -s > $O/widths.log 2>&1; echo "widths rc=$?" >> $O/host.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" > $O/attn_tests.log 2>&1; echo "attn rc=$?" >> $O/host.txt
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -s -k "forward_backward_parity" > $O/model.log 2>&1; echo "model rc=$?" >> $O/host.txt
rocprofv3 -L > $O/counters_full.txt 2>&1
for pm in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $pm | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pm -d $O/pmc_$tag -o pmc --output-format csv -- python3 tools/pmc_attn.py > $O/pmc_$tag.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats -d $O/attn_stats -o st --output-format csv -- python3 tools/pmc_attn.py > $O/attn_stats.log 2>&1
find $O -name "*.db" -delete; find $O -size +20M -delete
