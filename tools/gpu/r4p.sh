# round 4: stage-major GELU + GELU' epilogue arithmetic -- A/B on the up-projection forms (pp256a; compare with r4o's GM 4 rows of the same box class) + the whole GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4p; mkdir -p $O
timeout 600 python tools/bench_gemm_ab.py 5 pp256a,pp256x,pp256 gelu > $O/ab.log 2>&1
grep -v "amdgpu" $O/ab.log | cut -c1-250
timeout 3000 python -m pytest tests -m gpu -q -rf > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -10 | cut -c1-300
