cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ab; mkdir -p $O
timeout 900 python tools/check_packed_trajectory.py 40 16 > $O/traj.log 2>&1; grep -v amdgpu $O/traj.log | tail -45
