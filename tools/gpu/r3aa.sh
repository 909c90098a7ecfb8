cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3aa; mkdir -p $O
T=$PWD/$O/gemm_autotune_gfx950.json
cp profiles/gemm_autotune_gfx950.json $T
for extra in "--batch 3 --grad-accum 2" "--batch 3 --grad-accum 2 --fuse-accum" "--batch 6"; do
  UNIMP_GEMM_TUNE_FILE=$T UNIMP_GEMM_TUNE_WRITE=1 timeout 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline $extra > $O/tune.json 2> $O/tune.err
done
cp $T profiles/gemm_autotune_gfx950.json
timeout 900 python bench.py --steps 120 --warmup 5 --no-cpu-baseline > $O/soak_default.json 2> $O/soak_default.err
timeout 900 python bench.py --steps 120 --warmup 5 --no-cpu-baseline --packed > $O/soak_packed.json 2> $O/soak_packed.err
for f in $O/soak_*.json; do python -c "import json,sys; j=json.load(open('$f')); print('$f', j['value'], j['ms_per_step'], j['roofline']['frac'] if j.get('roofline') else None, j['config']['loss'], (j.get('packed_token_order') or {}).get('value'), j['config']['gemm_autotune'])"; done
