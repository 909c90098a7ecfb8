cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2r; mkdir -p $O
timeout 900 python -m pytest tests/test_dp_gpu.py -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/rc.txt
