# round 5: gemm8 (w8x = gemm7's loop at two waves per SIMD) -- bits and A/B against the family
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "w4x" > $O/pytest_gemm.log 2>&1; echo "pytest_gemm rc=$?" >> $O/rc.txt
tail -6 $O/pytest_gemm.log
timeout 1200 python tools/bench_gemm_ab.py 3 pp256a,w4x,w8x > $O/gemm_ab.log 2>&1; echo "gemm_ab rc=$?" >> $O/rc.txt
cat $O/gemm_ab.log; cat $O/rc.txt
