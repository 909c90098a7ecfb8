cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3am; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -rf -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -6
