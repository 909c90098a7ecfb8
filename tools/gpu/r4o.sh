# round 4: raster group height sweep of the whole-row-A kernel (tile rows per group; an XCD's 32 CUs run GM x 32/GM patches) + the reworked cfg5-width fp8 test
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4o; mkdir -p $O
for gm in 4 2 8 16; do
  UNIMP_GEMM_GM=$gm timeout 600 python tools/bench_gemm_ab.py 3 pp256a > $O/ab_gm$gm.log 2>&1
done
for gm in 4 2 8 16; do echo "== GM $gm"; grep -v "amdgpu\|^#" $O/ab_gm$gm.log | awk '{print $1,$2,$3,$4,$5,$6,$7,$8}' | cut -c1-100; done
timeout 900 python -m pytest "tests/test_widths_gpu.py::test_cfg5_width_fp8_loss_curve_against_the_chaos_floor" -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|cfg5 width|step-0|bf16|fp8 " $O/pytest.log | tail -8 | cut -c1-300
