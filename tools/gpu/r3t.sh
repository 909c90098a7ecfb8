cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3t; mkdir -p $O
timeout 900 python tools/hunt_invariance.py replay 1 > $O/hunt_replay.log 2>&1; grep -v "^    " $O/hunt_replay.log | cut -c1-300 | tail -8
timeout 900 python tools/hunt_invariance.py random 3 > $O/hunt_random.log 2>&1; grep -v "^    " $O/hunt_random.log | cut -c1-300 | tail -5
timeout 900 python tools/hunt_invariance.py random 2 9b > $O/hunt_random_9b.log 2>&1; grep -v "^    " $O/hunt_random_9b.log | cut -c1-300 | tail -4
timeout 2400 python -m pytest tests -m gpu -q -rf > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -10
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
python -c "import json; j=json.load(open('$O/bench_default.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['config']['gemm_autotune'], j['cpu_baseline']['value'])"
cat $O/rc.txt
