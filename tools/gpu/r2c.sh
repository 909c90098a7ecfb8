cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" > $O/attn_tests.log 2>&1; echo "attn tests rc=$?" >> $O/rc.txt
timeout 600 python tools/bench_attn2.py > $O/bench_attn2.log 2>&1; echo "bench_attn2 rc=$?" >> $O/rc.txt
timeout 1200 python -m pytest tests/test_dp_gpu.py tests/test_model_gpu.py tests/test_fullsize_gpu.py -q -s -k "rccl or two_rank or scheduler or checkpoint or beam or compact_head or forward_backward_parity or train_steps" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/rc.txt
# A/B on one box: head backward dense vs labeled rows only (committed autotune table, no live tuning)
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dense-head-backward > $O/bench_dense_head.json 2> $O/bench_dense_head.err
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_compact_head.json 2> $O/bench_compact_head.err
UNIMP_ATTN_GEN=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_attn_gen1.json 2> $O/bench_attn_gen1.err
echo done >> $O/rc.txt
