cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ag; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "stored_derivative or gemm" > $O/k.log 2>&1; tail -12 $O/k.log | cut -c1-300
# A/B with a table that gets the (possibly different) best variants of both forms: tune into two private copies
for mode in 0 1; do
  T=$PWD/$O/tune_$mode.json; cp profiles/gemm_autotune_gfx950.json $T
  python - <<PY
import json
t=json.load(open("$T"))
# drop the two epilogue classes whose cost changes (second output; stored-derivative input) for the LM / xattn MLP shapes: they are re-tuned live
keep={k:v for k,v in t.items() if not (json.loads(k)[5] in (True,"out2") and 10240 in json.loads(k)[:3])}
json.dump(keep,open("$T","w"))
print(len(t),len(keep))
PY
  UNIMP_DERIV_U8=$mode UNIMP_GEMM_TUNE_FILE=$T UNIMP_GEMM_TUNE_WRITE=1 timeout 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-packed-leg > $O/tune.json 2> $O/tune.err
done
for i in 1 2; do for mode in 0 1; do
  UNIMP_BENCH_SHAPES=1 UNIMP_DERIV_U8=$mode UNIMP_GEMM_TUNE_FILE=$PWD/$O/tune_$mode.json timeout 900 python bench.py --no-cpu-baseline --no-packed-leg > $O/bench_u8${mode}_$i.json 2> $O/bench_u8${mode}_$i.err
done; done
for f in $O/bench_*.json; do python -c "import json,sys; j=json.load(open('$f')); print('$f', j['value'], j['ms_per_step'], j['roofline']['frac'], j['config']['loss'], j['config']['gemm_autotune']['tuned_live_this_run'])"; done
grep "N= 10240\|K= 10240" $O/bench_u80_1.err | head -8; echo; grep "N= 10240\|K= 10240" $O/bench_u81_1.err | head -8
