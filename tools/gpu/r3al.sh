cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r3al; mkdir -p $O
for v in 4 8 9 5; do
python - <<PY
import json
t=json.load(open("profiles/gemm_autotune_gfx950.json"))
t['[32768, 10240, 2560, false, true, true]']=$v
json.dump(t,open("$O/tune_$v.json","w"))
PY
UNIMP_BENCH_SHAPES=1 UNIMP_GEMM_TUNE_FILE=$O/tune_$v.json timeout 900 python bench.py --no-cpu-baseline --no-packed-leg --steps 12 > $O/bench_$v.json 2> $O/bench_$v.err
python -c "import json; j=json.load(open('$O/bench_$v.json')); print($v, j['value'], j['ms_per_step'], j['roofline']['frac'])"
grep "M= 32768 N= 10240 K=  2560 aks=0 bks=1" $O/bench_$v.err
done
