#!/bin/bash
# round 6: the whole GPU suite on the current tree
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_g; mkdir -p $O
timeout 3300 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $O/rc.txt; tail -15 $O/pytest_gpu.log; cat $O/rc.txt
