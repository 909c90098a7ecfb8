cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2q; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "packed" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/rc.txt
timeout 600 python tools/bench_packed.py > $O/bench_packed.log 2>&1; echo "bench rc=$?" >> $O/rc.txt
