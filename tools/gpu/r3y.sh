cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3y; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "packed_rows" 2>&1 | tail -3
( time timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
python -c "import json; j=json.load(open('$O/bench_default.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['config']['gemm_autotune'], j['cpu_baseline']['value'], j.get('packed_token_order'))" | cut -c1-400
timeout 600 python -m pytest tests/test_dp_gpu.py -m gpu -q -k "bench" 2>&1 | tail -3
