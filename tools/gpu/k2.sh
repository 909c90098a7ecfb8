cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/k2; mkdir -p $O
cp profiles/gemm_autotune_gfx950.json $O/table.json
T=$PWD/$O/table.json
UNIMP_GEMM_TUNE_FILE=$T UNIMP_GEMM_TUNE_WRITE=1 UNIMP_BENCH_SHAPES=1 timeout 900 python bench.py --steps 10 --warmup 3 --cpu-full-steps 0 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
UNIMP_GEMM_TUNE_FILE=$T UNIMP_BENCH_SHAPES=1 timeout 900 python bench.py --steps 10 --warmup 3 --cpu-full-steps 0 > $O/bench2.json 2> $O/bench2.err; echo "bench2 rc=$?" >> $O/rc.txt
