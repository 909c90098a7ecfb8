cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3k; mkdir -p $O
for dp in 1 0; do echo "UNIMP_SKINNY_DEEP=$dp"; UNIMP_SKINNY_DEEP=$dp timeout 300 python tools/bench_skinny.py 10 2>&1 | grep "M="; done > $O/skinny_deep.log 2>&1; cat $O/skinny_deep.log
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k skinny > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for i in 1 2 3 4; do timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -x > $O/fullsize_$i.log 2>&1; tail -1 $O/fullsize_$i.log; grep -n "first module" $O/fullsize_$i.log | head -3; done
