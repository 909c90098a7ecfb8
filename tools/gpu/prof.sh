cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof.log 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
