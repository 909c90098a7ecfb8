#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
SKINNY_LN=1 timeout 300 python tools/bench_skinny.py 1 2>&1 | grep -v amdgpu
SKINNY_LN=1 timeout 300 python tools/bench_skinny.py 10 2>&1 | grep -v amdgpu
