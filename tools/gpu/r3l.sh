# round 3: evidence runs -- 2-rank self-launched bench over gloo on one GPU; soak; timed-steps window of the fused small-batch step
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3l; mkdir -p $O
UNIMP_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 6 --warmup 2 --batch 16 --no-cpu-baseline > $O/bench_gpus2_gloo_one_gpu.json 2> $O/bench_gpus2.err; echo "gpus2 rc=$?" > $O/rc.txt
wc -l $O/bench_gpus2_gloo_one_gpu.json; cut -c1-300 $O/bench_gpus2_gloo_one_gpu.json
timeout 900 python bench.py --steps 120 --warmup 3 --no-cpu-baseline --no-roofline > $O/soak120.json 2> $O/soak120.err; echo "soak rc=$?" >> $O/rc.txt
python -c "import json; j=json.load(open('$O/soak120.json')); print('soak', j['value'], j['ms_per_step'], j['config']['loss'])"
timeout 900 rocprofv3 --kernel-trace -d $O/trace -o tr --output-format csv -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --batch 3 --grad-accum 2 --fuse-accum > $O/trace.log 2>&1
python tools/trace_window.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 6 $O/r03_bench_b3ga2_fused_timed_steps.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cat $O/rc.txt
