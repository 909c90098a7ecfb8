cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3w; mkdir -p $O
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q -k "packed" > $O/m.log 2>&1; tail -12 $O/m.log | cut -c1-300
T=$PWD/$O/gemm_autotune_gfx950.json
cp profiles/gemm_autotune_gfx950.json $T
UNIMP_GEMM_TUNE_FILE=$T UNIMP_GEMM_TUNE_WRITE=1 timeout 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --packed --model 9b > $O/tune.json 2> $O/tune.err
cp $T profiles/gemm_autotune_gfx950.json
timeout 900 python bench.py --no-cpu-baseline --model 9b > $O/bench_9b.json 2> $O/bench_9b.err
timeout 900 python bench.py --no-cpu-baseline --model 9b --packed > $O/bench_9b_packed.json 2> $O/bench_9b_packed.err
for f in $O/bench_*.json; do python -c "import json,sys; j=json.load(open('$f')); print('$f', j['value'], j['ms_per_step'], j['roofline']['frac'] if j.get('roofline') else None, j['config'].get('gemm_autotune'))"; done
timeout 2400 python -m pytest tests -m gpu -q -rf > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -10
