cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2s; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "mx or dot or packed" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/rc.txt
UNIMP_MX_TILE=256 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "mx" > $O/tests256.log 2>&1; echo "tests256 rc=$?" >> $O/rc.txt
for t in 128 256; do echo "== tile $t" >> $O/bench_mx.log; UNIMP_MX_TILE=$t timeout 600 python tools/bench_mx.py 2>&1 | grep -v amdgpu >> $O/bench_mx.log; done
timeout 3000 python -m pytest tests -q -m gpu > $O/all_tests.log 2>&1; echo "all tests rc=$?" >> $O/rc.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
