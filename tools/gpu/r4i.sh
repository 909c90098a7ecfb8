# round 4 checkpoint: whole GPU suite, then re-tune the default bench's shapes from scratch and run the default bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4i; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -rf -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
T=$PWD/$O/gemm_autotune_gfx950.json; rm -f $T
UNIMP_GEMM_TUNE_FILE=$T UNIMP_GEMM_TUNE_WRITE=1 timeout 1200 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $O/tune.json 2> $O/tune.err; echo "tune rc=$?" >> $O/rc.txt
UNIMP_GEMM_TUNE_FILE=$T UNIMP_BENCH_SHAPES=1 timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
cat $O/rc.txt; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -8; python - <<'PY'
import json
j=json.load(open("gpurun_out/r4i/bench_default.json"))
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["gemm_ms_per_step"], "tuned live", j["config"]["gemm_autotune"])
print({k:(v.get("value"), v.get("roofline_frac")) if isinstance(v,dict) else v for k,v in (j.get("other_shapes") or {}).items()})
print("cfg5", j.get("cfg5_fp8"))
print("parity", {k:v for k,v in (j.get("parity") or {}).items() if k not in ("config","note")})
PY
grep "^  gemm M=" $O/bench_default.err | head -24
