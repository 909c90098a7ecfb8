cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2e; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" > $O/attn_tests.log 2>&1; echo "attn tests rc=$?" >> $O/rc.txt
timeout 600 python tools/bench_attn2.py > $O/bench_attn2.log 2>&1; echo "bench_attn2 rc=$?" >> $O/rc.txt
timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -s -k "beam" > $O/tests.log 2>&1; echo "beam rc=$?" >> $O/rc.txt
export ATTN=lm REP=2
for pm in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $pm | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pm -d $O/pmc_$tag -o pmc --output-format csv -- python3 tools/pmc_attn.py > $O/pmc_$tag.log 2>&1
done
find $O -name "*.db" -delete
