# round 3, sixth GPU pass: decode kernel profile; v1 spread A/B; full suite
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats -d $O/dec -o st --output-format csv -- python3 tools/bench_decode.py quick > $O/dec.log 2>&1
python - <<'PY' > $O/decode_kernels.txt
import csv, glob
f = glob.glob("gpurun_out/r3f/dec/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:28]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>7s} avg_us {float(r['AverageNs'])/1e3:9.2f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {r['Percentage']}%")
PY
cat $O/decode_kernels.txt
timeout 600 python bench.py --no-cpu-baseline --batch 3 --grad-accum 2 > $O/bench_b3ga2.json 2> $O/bench_b3ga2.err
UNIMP_V1_SPREAD=1 timeout 600 python bench.py --no-cpu-baseline --batch 3 --grad-accum 2 > $O/bench_b3ga2_spread.json 2> $O/bench_b3ga2_spread.err
UNIMP_V1_SPREAD=1 timeout 600 python bench.py --no-cpu-baseline --batch 3 --grad-accum 2 --fuse-accum > $O/bench_b3ga2_fused_spread.json 2> $O/bench_b3ga2_fused_spread.err
for f in $O/bench_b*.json; do python -c "import json,sys; j=json.load(open('$f')); print('$f', j['value'], j['ms_per_step'], j['roofline']['frac'] if j['roofline'] else None)"; done
timeout 2400 python -m pytest tests -m gpu -q -rf > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -10
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
