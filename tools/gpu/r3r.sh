cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3r; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -rf > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -10
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
python -c "import json; j=json.load(open('$O/bench_default.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['config']['gemm_autotune'], j['cpu_baseline']['value'])"
cat $O/rc.txt
