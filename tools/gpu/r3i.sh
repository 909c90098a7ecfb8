# round 3, ninth GPU pass: new skinny GEMM (tests + bandwidth + decode), batch-invariance diagnostic
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -q -rf -k "skinny or decode or gemm or generate or kv_cache or beam" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -10
for cfg in "" "8,1" "8,2" "8,4" "16,1" "16,2"; do
  echo "UNIMP_SKINNY_CFG=$cfg"; UNIMP_SKINNY_CFG=$cfg timeout 300 python tools/bench_skinny.py 10 2>&1 | grep "M=" 
  UNIMP_SKINNY_CFG=$cfg timeout 300 python tools/bench_skinny.py 40 2>&1 | grep "M="
done > $O/skinny.log 2>&1; cat $O/skinny.log
timeout 900 python tools/bench_decode.py quick > $O/decode.log 2>&1; grep "every beam\|training" $O/decode.log
timeout 900 python tools/debug_batch_invariance.py 9b 8 > $O/invariance_9b.log 2>&1; grep "trial" $O/invariance_9b.log
cat $O/rc.txt
