cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s; mkdir -p $O
timeout 900 python tools/hunt_invariance.py replay 1 > $O/hunt_replay.log 2>&1; grep -v "^    " $O/hunt_replay.log | cut -c1-420 | tail -40
