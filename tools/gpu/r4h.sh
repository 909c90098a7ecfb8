# round 4: does a persistent kernel gain once its tile-top wait stops covering the previous epilogue's stores? (rotary instantiation of gemm6x)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h; mkdir -p $O
timeout 900 python tools/bench_gemm_ab.py 5 pp256x,pp256px,pp256a,pp256p rope > $O/ab.log 2>&1; echo "ab rc=$?" > $O/rc.txt
UNIMP_MX_PP=0 timeout 300 python tools/bench_mx.py > $O/mx_lockstep.log 2>&1; echo "mx0 rc=$?" >> $O/rc.txt
timeout 300 python tools/bench_mx.py > $O/mx_pp.log 2>&1; echo "mx1 rc=$?" >> $O/rc.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "mx or rotary" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
cat $O/rc.txt; grep -v amdgpu.ids $O/ab.log; echo "== lockstep"; grep -v amdgpu.ids $O/mx_lockstep.log | cut -c1-200; echo "== ping-pong"; grep -v amdgpu.ids $O/mx_pp.log | cut -c1-200; tail -3 $O/pytest.log
