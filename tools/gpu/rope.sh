cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/rope; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "rope" > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 300 python tools/bench_rope.py > $O/new.log 2>&1
if [ -f tools/micro/libold.so ]; then
  cp unimp_amd/libunimp_hip.so /tmp/new.so; cp tools/micro/libold.so unimp_amd/libunimp_hip.so
  timeout 300 python tools/bench_rope.py > $O/old.log 2>&1
  cp /tmp/new.so unimp_amd/libunimp_hip.so
fi
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -x > $O/model.log 2>&1; echo "model rc=$?" >> $O/rc.txt
