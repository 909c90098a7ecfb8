# round 4: unit alpha / gate multiplies skipped + staged QuickGELU -- GEMM tests (bits across variants and epilogues), A/B, the model tests that compare bits
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4q; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm or derivative" > $O/pytest_gemm.log 2>&1; echo "pytest gemm rc=$?"
timeout 900 python tools/bench_gemm_ab.py 5 pp256a,pp256 > $O/ab.log 2>&1
grep -v amdgpu $O/ab.log | cut -c1-200
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_preprocess_gpu.py -m gpu -q -x > $O/pytest_model.log 2>&1; echo "pytest model rc=$?"
tail -3 $O/pytest_gemm.log; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_model.log | tail -5
