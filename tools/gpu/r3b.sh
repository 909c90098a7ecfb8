# round 3: full GPU suite, every failure listed
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -rf > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -60
