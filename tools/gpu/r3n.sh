cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3n; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -q -rf -s -k "packed" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "^\[packed|^(FAILED|ERROR)|passed|failed|Error" $O/pytest.log | tail -20
timeout 900 python bench.py --no-cpu-baseline --packed > $O/bench_packed.json 2> $O/bench_packed.err; echo "packed rc=$?" >> $O/rc.txt; tail -3 $O/bench_packed.err
timeout 900 python bench.py --no-cpu-baseline > $O/bench_padded.json 2> $O/bench_padded.err
for f in $O/bench_*.json; do python -c "import json,sys; j=json.load(open('$f')); print('$f', j['value'], j['ms_per_step'], j['roofline']['frac'] if j['roofline'] else None, j['config']['gemm_autotune'], j['config']['loss'])"; done
cat $O/rc.txt
