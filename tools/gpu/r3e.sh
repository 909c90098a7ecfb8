# round 3, fifth GPU pass: touched tests; small-batch benches with the dX split-K; power experiment; decode timings
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; mkdir -p $O
timeout 1800 python -m pytest tests/test_widths_gpu.py "tests/test_model_gpu.py::test_fused_accumulation_equals_sequential" tests/test_kernels_gpu.py -q -rf -s -k "widths or fused or gemm or splitk" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "^\[cfg5 fp8|^\[cfg4|^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -20
timeout 600 python bench.py --no-cpu-baseline --batch 3 --grad-accum 2 > $O/bench_b3ga2.json 2> $O/bench_b3ga2.err
timeout 600 python bench.py --no-cpu-baseline --batch 3 --grad-accum 2 --fuse-accum > $O/bench_b3ga2_fused.json 2> $O/bench_b3ga2_fused.err
UNIMP_BENCH_SHAPES=1 timeout 600 python bench.py --no-cpu-baseline --batch 6 --steps 8 > $O/bench_b6.json 2> $O/bench_b6.err
timeout 600 python bench.py --no-cpu-baseline --batch 16 > $O/bench_b16.json 2> $O/bench_b16.err
for f in $O/bench_b*.json; do python -c "import json,sys; j=json.load(open('$f')); print('$f', j['value'], j['ms_per_step'], j['roofline']['frac'] if j['roofline'] else None)"; done
grep "gemm M=" $O/bench_b6.err | head -16
timeout 300 python tools/bench_gemm_power.py > $O/gemm_power.log 2>&1; cat $O/gemm_power.log
timeout 900 python tools/bench_decode.py > $O/bench_decode.log 2>&1; cat $O/bench_decode.log
cat $O/rc.txt
