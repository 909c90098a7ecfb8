cd $GRAFT_REPO_ROOT
O=gpurun_out/r4l; mkdir -p $O
timeout 300 python tools/scratch/fp8curve.py > $O/curve.log 2>&1
UNIMP_FROZEN_WT=0 timeout 300 python tools/scratch/fp8curve.py > $O/curve_nowt.log 2>&1
grep -v amdgpu $O/curve.log | cut -c1-700; echo ==; grep -v amdgpu $O/curve_nowt.log | cut -c1-700
