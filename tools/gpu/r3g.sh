# round 3, seventh GPU pass: grouped decode attention tests + timings; GEMM variant bit identity; 9b invariance test rerun
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_preprocess_gpu.py -q -rf -k "decode or beam or generate or eval or kv_cache or cfg5_9b" --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "^(FAILED|ERROR)|passed|failed|s call" $O/pytest.log | tail -20
timeout 600 python tools/check_variant_bits.py > $O/variant_bits.log 2>&1; cat $O/variant_bits.log
timeout 900 python tools/bench_decode.py > $O/bench_decode.log 2>&1; cat $O/bench_decode.log
cat $O/rc.txt
