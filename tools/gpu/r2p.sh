cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2p; mkdir -p $O
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -s -k "fp8" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/rc.txt
timeout 900 python bench.py --model 9b --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_9b.json 2> $O/bench_9b.err; echo "9b rc=$?" >> $O/rc.txt
timeout 900 python bench.py --model 9b --fp8 --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_9b_fp8.json 2> $O/bench_9b_fp8.err; echo "9b fp8 rc=$?" >> $O/rc.txt
timeout 900 python bench.py --fp8 --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_4b_fp8.json 2> $O/bench_4b_fp8.err; echo "4b fp8 rc=$?" >> $O/rc.txt
