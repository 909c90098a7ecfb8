cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r3aj; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "stored_derivative or persistent or ragged" 2>&1 | tail -3
for i in 1 2 3; do
  ( cd build/ab_old && timeout 900 python bench.py --no-cpu-baseline --no-packed-leg > $O/bench_old_$i.json 2> $O/bench_old_$i.err )
  UNIMP_BENCH_SHAPES=1 timeout 900 python bench.py --no-cpu-baseline --no-packed-leg > $O/bench_new_$i.json 2> $O/bench_new_$i.err
done
for f in $O/bench_*.json; do python -c "import json,sys; j=json.load(open('$f')); print('$f', j['value'], j['ms_per_step'], j['roofline']['frac'], j['config']['loss'], j['config']['gemm_autotune']['tuned_live_this_run'])"; done
grep "N= 10240\|K= 10240" $O/bench_new_1.err | head -6
