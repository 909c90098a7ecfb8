cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3o; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -q -rf -s -k "packed" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "^\[packed|^(FAILED|ERROR)|passed|failed|^E " $O/pytest.log | tail -12
T=$PWD/$O/tune.json; cp profiles/gemm_autotune_gfx950.json $T
UNIMP_GEMM_TUNE_FILE=$T UNIMP_GEMM_TUNE_WRITE=1 timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline --packed > $O/tune_packed.json 2> $O/tune_packed.err
UNIMP_GEMM_TUNE_FILE=$T timeout 900 python bench.py --no-cpu-baseline --packed > $O/bench_packed.json 2> $O/bench_packed.err
UNIMP_GEMM_TUNE_FILE=$T UNIMP_BENCH_SHAPES=1 timeout 900 python bench.py --no-cpu-baseline > $O/bench_padded.json 2> $O/bench_padded.err
for f in $O/bench_*.json; do python -c "import json,sys; j=json.load(open('$f')); print('$f', j['value'], j['ms_per_step'], j['roofline']['frac'] if j['roofline'] else None, j['config']['gemm_autotune'], j['config']['loss'])"; done
timeout 900 rocprofv3 --kernel-trace -d $O/trace -o tr --output-format csv -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --packed > $O/trace.log 2>&1
UNIMP_GEMM_TUNE_FILE=$T python tools/trace_window.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 6 $O/r03_bench_b64_packed_timed_steps.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cat $O/rc.txt
