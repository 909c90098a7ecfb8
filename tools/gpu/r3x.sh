cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3x; mkdir -p $O
T=$PWD/$O/gemm_autotune_gfx950.json
cp profiles/gemm_autotune_gfx950.json $T
for extra in "--model 9b" "--batch 3 --grad-accum 2 --fuse-accum" "--batch 16" "--batch 32" "--model 9b --task img_gen --batch 12"; do
  UNIMP_GEMM_TUNE_FILE=$T UNIMP_GEMM_TUNE_WRITE=1 timeout 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --packed $extra > $O/tune.json 2> $O/tune.err
done
cp $T profiles/gemm_autotune_gfx950.json
i=0
for extra in "--model 9b" "--batch 3 --grad-accum 2 --fuse-accum" "--batch 16" "--batch 32" "--model 9b --task img_gen --batch 12"; do
  i=$((i+1))
  timeout 900 python bench.py --no-cpu-baseline $extra > $O/bench_${i}_padded.json 2> $O/bench_${i}_padded.err
  timeout 900 python bench.py --no-cpu-baseline --packed $extra > $O/bench_${i}_packed.json 2> $O/bench_${i}_packed.err
done
for f in $O/bench_*.json; do python -c "import json,sys; j=json.load(open('$f')); print('$f', j['value'], j['ms_per_step'], j['config'].get('gemm_autotune'))"; done
