# round 3, fourth GPU pass: the tests touched since r3c; b table; fused accumulation; fp8 tests
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O
timeout 1500 python -m pytest tests/test_widths_gpu.py tests/test_preprocess_gpu.py "tests/test_model_gpu.py::test_fused_accumulation_equals_sequential" "tests/test_model_gpu.py::test_fp8_loss_curve_tracks_bf16" "tests/test_model_gpu.py::test_graphed_micro_step_equals_eager" tests/test_kernels_gpu.py::test_split_key_decode_attention -q -rf -s > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "^\[|^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -40
timeout 600 python bench.py --no-cpu-baseline --batch 3 --grad-accum 2 --fuse-accum > $O/bench_b3ga2_fused.json 2> $O/bench_b3ga2_fused.err; echo "fused rc=$?" >> $O/rc.txt
timeout 600 python bench.py --no-cpu-baseline --batch 6 > $O/bench_b6.json 2> $O/bench_b6.err
timeout 600 python bench.py --no-cpu-baseline --batch 32 > $O/bench_b32.json 2> $O/bench_b32.err
for f in $O/bench_b*.json; do python -c "import json,sys; j=json.load(open('$f')); print('$f', j['value'], j['ms_per_step'], j['roofline']['frac'] if j['roofline'] else None)"; done
UNIMP_BENCH_SHAPES=1 timeout 600 python bench.py --no-cpu-baseline --batch 3 --grad-accum 2 --steps 8 > $O/shapes_b3.json 2> $O/shapes_b3.err
grep "gemm M=" $O/shapes_b3.err | head -40
timeout 300 python tools/bench_gemm_power.py > $O/gemm_power.log 2>&1; cat $O/gemm_power.log
cat $O/rc.txt
