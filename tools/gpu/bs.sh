cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/bs; mkdir -p $O
for b in 128 96; do
timeout 1200 python bench.py --batch $b --steps 6 --warmup 3 --cpu-full-steps 0 > $O/bench_b$b.json 2> $O/bench_b$b.err; echo "b$b rc=$?" >> $O/rc.txt
done
