# round 4: the PMC leg with the round-4 kernel instances + the tests that failed / are new since the final run
cd $GRAFT_REPO_ROOT
cp -r profiles /tmp/profiles_before
ROUND=r04 bash tools/gpu/final.sh pmc > gpurun_out/final_pmc.log 2>&1
mkdir -p gpurun_out/final_pmc_profiles; for f in profiles/r04_pmc_gemm.csv profiles/r04_pmc_gemm.json profiles/r04_pmc_attention.csv; do [ -f $f ] && cp $f gpurun_out/final_pmc_profiles/; done
cat gpurun_out/final/rc.txt | tail -20
timeout 1500 python -m pytest "tests/test_model_gpu.py::test_graphed_micro_step_equals_eager" "tests/test_model_gpu.py::test_fp8_loss_curve_tracks_bf16" "tests/test_widths_gpu.py::test_cfg5_width_fp8_loss_curve_against_the_chaos_floor" -q -s > gpurun_out/r4m_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|fp8 loss|cfg5 width" gpurun_out/r4m_pytest.log | tail -8 | cut -c1-400
cat gpurun_out/final/pmc_to_json.err 2>/dev/null | tail -5
