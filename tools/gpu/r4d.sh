# round 4: k-strided (transposed copy) weight operand in forward GEMMs: bits across variants + A/B against the k-contiguous form
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
timeout 600 python tools/check_ks_bits.py > $O/ks_bits.log 2>&1; echo "ks_bits rc=$?" > $O/rc.txt
timeout 1500 python tools/bench_gemm_ab.py 5 pp256,pp256x,w8 > $O/ab.log 2>&1; echo "ab rc=$?" >> $O/rc.txt
grep -v amdgpu.ids $O/ks_bits.log; grep -v amdgpu.ids $O/ab.log
