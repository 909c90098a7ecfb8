# round 3, first GPU pass: full GPU suite (all failures listed), default bench, 1-rank RCCL path, RCCL-footprint interference
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_dp_gpu.py::test_bench_gpus2_starts_its_own_ranks > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 900 python -m pytest tests/test_dp_gpu.py::test_bench_gpus2_starts_its_own_ranks -q -x > $O/pytest_launcher.log 2>&1; echo "launcher rc=$?" >> $O/rc.txt
UNIMP_BENCH_SHAPES=1 timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
timeout 900 python bench.py --no-cpu-baseline --dp-hooks > $O/bench_dphooks.json 2> $O/bench_dphooks.err; echo "dphooks rc=$?" >> $O/rc.txt
timeout 600 python tools/bench_interference.py > $O/interference.log 2>&1; echo "interf rc=$?" >> $O/rc.txt
tail -3 $O/pytest.log; cat $O/rc.txt; cat $O/bench_default.json | cut -c1-400
