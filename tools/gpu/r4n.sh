cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4n
timeout 600 python tools/scratch/fp8curve_cfg5.py 2e-4 > gpurun_out/r4n/c1.log 2>&1
timeout 600 python tools/scratch/fp8curve_cfg5.py 5e-5 > gpurun_out/r4n/c2.log 2>&1
grep -v amdgpu gpurun_out/r4n/c1.log | cut -c1-400; grep -v amdgpu gpurun_out/r4n/c2.log | cut -c1-400
