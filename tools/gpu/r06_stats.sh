#!/bin/bash
# kernel stats of the default step on the final tree (the command final.sh's prof leg runs, without its other legs)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_stats; mkdir -p $O
timeout 500 rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-packed-leg --no-shape-legs --no-cfg5-leg --no-parity --no-decode-leg > $O/prof.log 2>&1
echo "rc=$?" > $O/rc.txt
f=$(find $O/stats -name '*kernel_stats.csv' | head -1); cp "$f" $O/kernel_stats.csv
tail -1 $O/prof.log | cut -c1-300; head -6 $O/kernel_stats.csv | cut -c1-150
find $O -name "*.db" -delete; find $O/stats -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
