cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2l; mkdir -p $O
UNIMP_BENCH_SHAPES=1 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
timeout 900 rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof.log 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_widths_gpu.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/rc.txt
