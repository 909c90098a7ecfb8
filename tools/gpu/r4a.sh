# round 4, first GPU pass: new kernel variants (one fragment set) -- correctness tests of the GEMM family + interleaved A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm or derivative" > $O/pytest_gemm.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 1200 python tools/bench_gemm_ab.py 5 > $O/ab.log 2>&1; echo "ab rc=$?" >> $O/rc.txt
tail -5 $O/pytest_gemm.log; cat $O/ab.log
