#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_n; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "dw_two or w4x_equals or variants_same_bits" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt; tail -15 $O/pytest.log; cat $O/rc.txt
