# round 4: fixed-epilogue-kind kernel instantiations (A/B by environment switch, two alternations), GEMM tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm or derivative" > $O/pytest_gemm.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
for i in 1 2; do
  UNIMP_GEMM_FIXED_EPI=0 timeout 900 python tools/bench_gemm_ab.py 3 pp256,pp256x "" > $O/ab_generic_$i.log 2>&1
  timeout 900 python tools/bench_gemm_ab.py 3 pp256,pp256x "" > $O/ab_fixed_$i.log 2>&1
done
tail -3 $O/pytest_gemm.log
for f in $O/ab_generic_1.log $O/ab_fixed_1.log $O/ab_generic_2.log $O/ab_fixed_2.log; do echo "== $f"; grep -v "amdgpu.ids" $f | cut -c1-200; done
