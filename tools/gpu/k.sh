cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/k; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "gemm" -x > $O/kern.log 2>&1; echo "kern rc=$?" > $O/rc.txt
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -x > $O/model.log 2>&1; echo "model rc=$?" >> $O/rc.txt
UNIMP_BENCH_SHAPES=1 timeout 900 python bench.py --steps 10 --warmup 3 --cpu-full-steps 0 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
