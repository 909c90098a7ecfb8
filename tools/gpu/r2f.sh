cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${OUT:-r2f}; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" > $O/attn_tests.log 2>&1; echo "attn tests rc=$?" >> $O/rc.txt
timeout 600 python tools/bench_attn2.py > $O/bench_attn2.log 2>&1; echo "bench_attn2 rc=$?" >> $O/rc.txt
