cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j; mkdir -p $O
timeout 900 python tools/debug_batch_invariance.py 9b 10 > $O/invariance_9b.log 2>&1; grep "trial" $O/invariance_9b.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -rf -k "skinny" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt; tail -2 $O/pytest.log
timeout 900 python tools/bench_decode.py quick > $O/decode.log 2>&1; grep "every beam\|training" $O/decode.log
