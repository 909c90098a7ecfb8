cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/soak; mkdir -p $O
timeout 1500 python bench.py --steps 120 --warmup 3 --cpu-full-steps 0 --no-roofline > $O/bench.json 2> $O/bench.err; echo "soak rc=$?" > $O/rc.txt
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch 3 --grad-accum 2 > $O/b3.json 2> $O/b3.err; echo "b3 rc=$?" >> $O/rc.txt
