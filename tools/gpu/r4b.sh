# round 4: L2 warm-up of the epilogue inputs in the one-set ping-pong kernel -- A/B against the unchanged two-set kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm or derivative" > $O/pytest_gemm.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 1200 python tools/bench_gemm_ab.py 5 pp256,pp256x,pp128,pp128x > $O/ab.log 2>&1; echo "ab rc=$?" >> $O/rc.txt
tail -3 $O/pytest_gemm.log; cat $O/ab.log
