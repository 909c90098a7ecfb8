cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3v; mkdir -p $O
T=$PWD/$O/gemm_autotune_gfx950.json
cp profiles/gemm_autotune_gfx950.json $T
UNIMP_GEMM_TUNE_FILE=$T UNIMP_GEMM_TUNE_WRITE=1 timeout 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --packed > $O/tune.json 2> $O/tune.err
cp $T profiles/gemm_autotune_gfx950.json
timeout 900 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
timeout 900 python bench.py --no-cpu-baseline --packed > $O/bench_packed.json 2> $O/bench_packed.err
timeout 900 python bench.py --no-cpu-baseline > $O/bench_default2.json 2> $O/bench_default2.err
timeout 900 python bench.py --no-cpu-baseline --packed > $O/bench_packed2.json 2> $O/bench_packed2.err
for f in $O/bench_*.json; do python -c "import json,sys; j=json.load(open('$f')); print('$f', j['value'], j['ms_per_step'], j['roofline']['frac'] if j.get('roofline') else None, j['config'].get('gemm_autotune'))"; done
timeout 900 rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --packed > $O/prof.log 2>&1
python tools/trace_window.py $(find $O/stats -name "*kernel_trace.csv" | head -1) 6 $O/r03_bench_b64_packed_timed_steps.csv > $O/window.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
head -12 $O/window.txt
