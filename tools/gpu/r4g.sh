# round 4: whole-row A staging (pp256a / pp128a): GEMM tests, bits with k-strided weights, A/B against the one-set kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm or derivative" > $O/pytest_gemm.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 600 python tools/check_ks_bits.py > $O/ks_bits.log 2>&1; echo "ks_bits rc=$?" >> $O/rc.txt
timeout 1500 python tools/bench_gemm_ab.py 5 pp256x,pp256a,pp256 > $O/ab.log 2>&1; echo "ab rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -5 $O/pytest_gemm.log; grep -v amdgpu.ids $O/ks_bits.log | cut -c1-260 | tail -6; grep -v amdgpu.ids $O/ab.log
