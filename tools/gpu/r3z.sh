cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3z; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -rf > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -10
timeout 900 python bench.py --no-cpu-baseline --batch 3 --grad-accum 2 > $O/bench_b3ga2.json 2> $O/bench_b3ga2.err
timeout 900 python bench.py --no-cpu-baseline --batch 3 --grad-accum 2 --fuse-accum > $O/bench_b3ga2_fused.json 2> $O/bench_b3ga2_fused.err
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for f in $O/bench_*.json; do python -c "import json,sys; j=json.load(open('$f')); print('$f', j['value'], j['ms_per_step'], j['roofline']['frac'] if j.get('roofline') else None, 'packed leg:', (j.get('packed_token_order') or {}).get('value'), j['config']['gemm_autotune'])"; done
timeout 900 python tools/bench_decode.py > $O/bench_decode.log 2>&1; tail -12 $O/bench_decode.log
