cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2n; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "rope or layernorm or norm" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/rc.txt
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
for b in 56 63 64; do timeout 900 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --batch $b > $O/bench_b$b.json 2> $O/bench_b$b.err; done
