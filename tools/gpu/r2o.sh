cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2o; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "mx" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/rc.txt
timeout 600 python tools/bench_mx.py > $O/bench_mx.log 2>&1; echo "bench rc=$?" >> $O/rc.txt
