# end-of-round measurement on the final tree: autotune table, bench lines, rocprofv3 kernel stats (whole process + timed steps only),
# PMC passes with pinned kernel instances.  Writes gpurun_out/final/; the summaries to keep are copied to profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${ROUND:-r03}
O=gpurun_out/final; mkdir -p $O
T=$PWD/$O/gemm_autotune_gfx950.json
cp profiles/gemm_autotune_gfx950.json $T        # keep the committed choices; only shapes / epilogue classes that are missing get tuned
# 1. autotune table for the shapes of the default bench (b = 64), the reference's shipped shape (b = 3, GA 2; fused: b = 6), b = 16 / 32 / 48, the 9b model
for extra in "" "--packed" "--batch 3 --grad-accum 2" "--batch 3 --grad-accum 2 --fuse-accum" "--batch 16" "--batch 32" "--batch 48" "--model 9b" "--model 9b --packed" "--model 9b --task img_gen --batch 12"; do
  UNIMP_GEMM_TUNE_FILE=$T UNIMP_GEMM_TUNE_WRITE=1 timeout 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline $extra > $O/tune.json 2> $O/tune.err
done
cp $T profiles/gemm_autotune_gfx950.json
# 2. bench lines with the table (no live tuning)
UNIMP_BENCH_SHAPES=1 timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?" >> $O/rc.txt
timeout 900 python bench.py --no-cpu-baseline --dp-hooks > $O/bench_dphooks.json 2> $O/bench_dphooks.err
timeout 900 python bench.py --no-cpu-baseline --packed > $O/bench_packed.json 2> $O/bench_packed.err
timeout 900 python bench.py --no-cpu-baseline --batch 3 --grad-accum 2 > $O/bench_b3ga2.json 2> $O/bench_b3ga2.err
timeout 900 python bench.py --no-cpu-baseline --batch 3 --grad-accum 2 --fuse-accum > $O/bench_b3ga2_fused.json 2> $O/bench_b3ga2_fused.err
timeout 900 python bench.py --no-cpu-baseline --batch 16 > $O/bench_b16.json 2> $O/bench_b16.err
timeout 900 python bench.py --no-cpu-baseline --batch 32 > $O/bench_b32.json 2> $O/bench_b32.err
timeout 900 python bench.py --no-cpu-baseline --batch 48 > $O/bench_b48.json 2> $O/bench_b48.err
timeout 900 python bench.py --no-cpu-baseline --model 9b > $O/bench_9b.json 2> $O/bench_9b.err
timeout 900 python bench.py --no-cpu-baseline --model 9b --fp8 > $O/bench_9b_fp8.json 2> $O/bench_9b_fp8.err
# BASELINE config 5's own workload: the 9b model on image-token generation samples (L = 1024, 2 history images, 257 labeled positions), bf16 and fp8
timeout 900 python bench.py --no-cpu-baseline --model 9b --task img_gen --batch 12 > $O/bench_9b_imggen.json 2> $O/bench_9b_imggen.err
timeout 900 python bench.py --no-cpu-baseline --model 9b --task img_gen --batch 12 --fp8 > $O/bench_9b_imggen_fp8.json 2> $O/bench_9b_imggen_fp8.err
for f in $O/bench_*.json; do python -c "import json,sys; j=json.load(open('$f')); print('$f', j['value'], j['ms_per_step'], j['roofline']['frac'] if j.get('roofline') else None, 'packed leg:', (j.get('packed_token_order') or {}).get('value'))"; done > $O/summary.txt 2>&1
# 3. kernel stats of the default bench command: whole process (--stats) and the timed steps only (markers)
timeout 900 rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-packed-leg > $O/prof.log 2>&1
python tools/trace_window.py $(find $O/stats -name "*kernel_trace.csv" | head -1) 6 $O/${R}_bench_b64_timed_steps.csv > $O/window.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $O/stats_packed -o st --output-format csv -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --packed > $O/prof_packed.log 2>&1
python tools/trace_window.py $(find $O/stats_packed -name "*kernel_trace.csv" | head -1) 6 $O/${R}_bench_b64_packed_timed_steps.csv > $O/window_packed.txt 2>&1
# 3b. micro-benchmarks of the HBM-bound kernels, streaming from HBM; decode
timeout 300 python tools/bench_ln.py --rotate 3 > $O/bench_ln.log 2>&1
timeout 300 python tools/bench_adamw.py > $O/bench_adamw.log 2>&1
timeout 600 python tools/bench_decode.py > $O/bench_decode.log 2>&1
timeout 300 python tools/bench_gemm_power.py > $O/gemm_power.log 2>&1
timeout 600 python tools/check_variant_bits.py > $O/variant_bits.log 2>&1
timeout 600 python tools/hunt_invariance.py random 2 > $O/hunt_random.log 2>&1
timeout 300 python tools/bench_skinny.py 10 > $O/skinny_m10.log 2>&1; timeout 300 python tools/bench_skinny.py 40 > $O/skinny_m40.log 2>&1
# 4. PMC passes: the step's dominant GEMM kernel instances, variants pinned
export PMC_MANIFEST=$PWD/$O/pmc_manifest.json
for pm in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $pm | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pm -d $O/pmcg_$tag -o pmc --output-format csv -- python3 tools/pmc_gemm_step.py > $O/pmcg_$tag.log 2>&1
done
python tools/pmc_to_json.py $O/pmc_manifest.json $O/pmcg_FETCH_SIZE $O/pmcg_WRITE_SIZE $O/pmcg_SQ_VALU_MFMA_BUSY_CYCLES $O/${R}_pmc_gemm > $O/pmc_rows.json 2> $O/pmc_to_json.err
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
echo done >> $O/rc.txt
cat $O/summary.txt $O/window.txt
# 5. the whole GPU suite on the final tree
timeout 2400 python -m pytest tests -m gpu -q -rf > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -10
