# round 4: register-direct epilogue of the fixed-kind kernels (pp256d) -- GEMM tests, bits, A/B against pp256a
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4j; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm or derivative" > $O/pytest_gemm.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 600 python tools/check_ks_bits.py > $O/ks_bits.log 2>&1; echo "ks_bits rc=$?" >> $O/rc.txt
timeout 1500 python tools/bench_gemm_ab.py 5 pp256a,pp256d,pp256x > $O/ab.log 2>&1; echo "ab rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -5 $O/pytest_gemm.log; grep -v amdgpu.ids $O/ks_bits.log | cut -c1-300 | tail -4; grep -v amdgpu.ids $O/ab.log
