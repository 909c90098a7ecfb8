# round 5, second GPU pass: gemm7 schedule 2 + gemm3b (peeled) -- bits, A/B, stamps; then the tests fixed after pass one
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm" > $O/pytest_gemm.log 2>&1; echo "pytest_gemm rc=$?" >> $O/rc.txt
tail -8 $O/pytest_gemm.log
timeout 1200 python tools/bench_gemm_ab.py 5 pp256a,pp256b,w4x,w4x_s1 > $O/gemm_ab.log 2>&1; echo "gemm_ab rc=$?" >> $O/rc.txt
cat $O/gemm_ab.log
for shp in "32768 2560 10240 plain 1" "32768 10240 2560 gelu2 1" "32768 2560 2560 res 1" "131584 1024 4096 res 1"; do
  timeout 300 python tools/stamp_gemm7.py $shp >> $O/stamps.log 2>&1
  set -- $shp
  timeout 300 python tools/stamp_gemm3.py $1 $2 $3 pp256a $4 $5 >> $O/stamps.log 2>&1
done
echo "stamps rc=$?" >> $O/rc.txt
cat $O/stamps.log
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py -m gpu -q -x -k "fused or flush or overlapped or sharded_optimizer_state_eight" > $O/pytest_a.log 2>&1; echo "pytest_a rc=$?" >> $O/rc.txt
tail -5 $O/pytest_a.log
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -k "full_depth" -s > $O/pytest_b.log 2>&1; echo "pytest_b rc=$?" >> $O/rc.txt
grep "cfg2 full depth" $O/pytest_b.log | head -1 | cut -c1-2500; tail -3 $O/pytest_b.log
cat $O/rc.txt
