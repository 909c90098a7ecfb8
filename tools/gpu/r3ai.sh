cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ai; mkdir -p $O
timeout 900 python tests/deriv_u8_gradients.py CFG2_SLIM > $O/grads.log 2>&1; grep -v amdgpu $O/grads.log | tail -8
