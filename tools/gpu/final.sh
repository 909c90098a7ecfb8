# end-of-round measurement on the final tree: autotune table, bench lines, rocprofv3 kernel stats, PMC passes.  Writes gpurun_out/final/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
T=$PWD/$O/gemm_autotune_gfx950.json
cp profiles/gemm_autotune_gfx950.json $T        # keep the committed choices; only shapes / epilogue classes that are missing get tuned
# 1. autotune table for the shapes of the default bench (b = 64), the reference's shipped shape (b = 3, GA 2) and the 9b model
UNIMP_GEMM_TUNE_FILE=$T UNIMP_GEMM_TUNE_WRITE=1 timeout 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $O/tune1.json 2> $O/tune1.err
UNIMP_GEMM_TUNE_FILE=$T UNIMP_GEMM_TUNE_WRITE=1 timeout 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --batch 3 --grad-accum 2 > $O/tune2.json 2> $O/tune2.err
UNIMP_GEMM_TUNE_FILE=$T UNIMP_GEMM_TUNE_WRITE=1 timeout 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --model 9b > $O/tune3.json 2> $O/tune3.err
UNIMP_GEMM_TUNE_FILE=$T UNIMP_GEMM_TUNE_WRITE=1 timeout 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --batch 48 > $O/tune4.json 2> $O/tune4.err
cp $T profiles/gemm_autotune_gfx950.json
# 2. bench lines with the table (no live tuning)
UNIMP_BENCH_SHAPES=1 timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?" >> $O/rc.txt
timeout 900 python bench.py --no-cpu-baseline --batch 3 --grad-accum 2 > $O/bench_b3ga2.json 2> $O/bench_b3ga2.err
timeout 900 python bench.py --no-cpu-baseline --model 9b > $O/bench_9b.json 2> $O/bench_9b.err
timeout 900 python bench.py --no-cpu-baseline --model 9b --fp8 > $O/bench_9b_fp8.json 2> $O/bench_9b_fp8.err
timeout 900 python bench.py --no-cpu-baseline --batch 48 > $O/bench_b48.json 2> $O/bench_b48.err
# 3. kernel stats of the default bench command
timeout 900 rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $O/stats48 -o st --output-format csv -- python3 bench.py --batch 48 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof48.log 2>&1
# 3b. micro-benchmarks of the HBM-bound kernels, streaming from HBM
timeout 300 python tools/bench_ln.py --rotate 3 > $O/bench_ln.log 2>&1
timeout 300 python tools/bench_rope.py > $O/bench_rope.log 2>&1
timeout 300 python tools/bench_adamw.py > $O/bench_adamw.log 2>&1
# 4. PMC passes: dominant GEMM shapes, attention kernels (both generations)
for pm in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $pm | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pm -d $O/pmcg_$tag -o pmc --output-format csv -- python3 tools/pmc_gemm_step.py > $O/pmcg_$tag.log 2>&1
done
export REP=2
for gen in 1 2; do
  for pm in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    tag=$(echo $pm | cut -d' ' -f1)
    UNIMP_ATTN_GEN=$gen timeout 600 rocprofv3 --kernel-trace --pmc $pm -d $O/pmca${gen}_$tag -o pmc --output-format csv -- python3 tools/pmc_attn.py > $O/pmca${gen}_$tag.log 2>&1
  done
done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
python tools/pmc_to_json.py $O/pmcg_* $O/r02_pmc_gemm 32768 > $O/pmc_gemm_rows.json 2> $O/pmc_to_json.err
python tools/pmc_summary.py $O/pmca1_* --match attn > $O/r02_pmc_attention_gen1.csv 2>> $O/pmc_to_json.err
python tools/pmc_summary.py $O/pmca2_* --match attn > $O/r02_pmc_attention_gen2.csv 2>> $O/pmc_to_json.err
echo done >> $O/rc.txt
