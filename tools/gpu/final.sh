# End-of-round measurement on the final tree.  Every leg records its exit code in $O/rc.txt; a summary reaches profiles/ ONLY through
# keep(), which refuses the copy when the leg failed, the file is empty or it holds a Python traceback (a crash log had
# been copied by hand and cited as a measurement).  Writes gpurun_out/final/; profiles/ files of this round are (re)written here.
#   usage: ROUND=r04 bash tools/gpu/final.sh [legs...]      legs: tune bench prof micro pmc pytest (default: all)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${ROUND:-r06}
O=gpurun_out/final; mkdir -p $O profiles
LEGS="${@:-tune bench prof micro pmc pytest}"
echo "# final.sh legs: $LEGS ($(date -u +%FT%TZ))" >> $O/rc.txt
declare -A RC
leg() {            # leg <name> <timeout> <stdout file> <command...>: run, record rc
  local name=$1 to=$2 out=$3; shift 3
  timeout $to "$@" > $out 2> ${out%.*}.err; RC[$name]=$?
  echo "$name rc=${RC[$name]}" >> $O/rc.txt
}
keep() {           # keep <leg name> <src> <dst under profiles/>: copy a summary only when its leg succeeded and the file is a measurement
  local name=$1 src=$2 dst=profiles/$3
  if [ "${RC[$name]}" != "0" ]; then echo "keep: $3 NOT copied (leg $name rc=${RC[$name]})" >> $O/rc.txt; return 1; fi
  if [ ! -s "$src" ]; then echo "keep: $3 NOT copied ($src empty or missing)" >> $O/rc.txt; return 1; fi
  if grep -q "Traceback (most recent call last)" "$src"; then echo "keep: $3 NOT copied ($src holds a traceback)" >> $O/rc.txt; return 1; fi
  cp "$src" "$dst"; mkdir -p $O/profiles; cp "$src" "$O/profiles/$3"       # profiles/ on the box is not merged back: the mirror under gpurun_out/ is
  echo "keep: $3 <- $src" >> $O/rc.txt
}
has() { case " $LEGS " in *" $1 "*) return 0;; *) return 1;; esac; }
T=$PWD/$O/gemm_autotune_gfx950.json

if has tune; then
  # 1. autotune table, re-tuned from scratch on this tree (new kernel variants, frozen weights read as W^T): the shapes of the default bench
  #    (b = 64 + its packed / b = 16 / b = 32 / reference-shape legs), b = 48, the sequential b = 3 x GA 2 form, the 9b model and its legs
  rm -f $T
  i=0
  for extra in "" "--packed --no-shape-legs" "--batch 3 --grad-accum 2" "--batch 3 --grad-accum 2 --no-fuse-accum" "--batch 48" "--model 9b" "--model 9b --packed" "--model 9b --task img_gen --batch 12" "--model 9b --fp8" "--model 9b --task img_gen --batch 12 --fp8"; do
    i=$((i+1))
    UNIMP_GEMM_TUNE_FILE=$T UNIMP_GEMM_TUNE_WRITE=1 leg tune$i 900 $O/tune$i.json python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline $extra
  done
  RC[tune]=0; for k in $(seq 1 $i); do [ "${RC[tune$k]}" = "0" ] || RC[tune]=1; done
  keep tune $T gemm_autotune_gfx950.json
fi

if has bench; then
  # 2. bench lines with the committed table (no live tuning expected: config.gemm_autotune.tuned_live_this_run)
  # the default command also completes the table (the parity leg's b = 1 shapes are not part of any tune run): written back and kept
  UNIMP_GEMM_TUNE_WRITE=1 UNIMP_BENCH_SHAPES=1 leg bench_default 1500 $O/bench_default.json python bench.py
  cp unimp_amd/gemm_autotune_gfx950.json $O/gemm_autotune_gfx950.json; keep bench_default $O/gemm_autotune_gfx950.json gemm_autotune_gfx950.json      # round 6: the table ships inside the package; profiles/ keeps the measured copy
  grep "^  gemm M=" $O/bench_default.err > $O/gemm_shapes.txt
  keep bench_default $O/gemm_shapes.txt ${R}_gemm_shapes_b64.txt
  leg bench_dphooks 900 $O/bench_dphooks.json python bench.py --no-cpu-baseline --dp-hooks
  leg bench_packed 900 $O/bench_packed.json python bench.py --no-cpu-baseline --packed
  leg bench_b3ga2 900 $O/bench_b3ga2.json python bench.py --no-cpu-baseline --batch 3 --grad-accum 2
  leg bench_b3ga2_dphooks 900 $O/bench_b3ga2_dphooks.json python bench.py --no-cpu-baseline --batch 3 --grad-accum 2 --dp-hooks
  leg bench_b3ga2_seq 900 $O/bench_b3ga2_seq.json python bench.py --no-cpu-baseline --batch 3 --grad-accum 2 --no-fuse-accum
  leg bench_b48 900 $O/bench_b48.json python bench.py --no-cpu-baseline --batch 48
  leg bench_overlap 900 $O/bench_overlap.json python bench.py --no-cpu-baseline --overlap-optimizer --no-shape-legs --no-cfg5-leg --no-packed-leg
  leg bench_9b 900 $O/bench_9b.json python bench.py --no-cpu-baseline --model 9b
  leg bench_9b_fp8 900 $O/bench_9b_fp8.json python bench.py --no-cpu-baseline --model 9b --fp8
  leg bench_9b_imggen 900 $O/bench_9b_imggen.json python bench.py --no-cpu-baseline --model 9b --task img_gen --batch 12
  leg bench_9b_imggen_fp8 900 $O/bench_9b_imggen_fp8.json python bench.py --no-cpu-baseline --model 9b --task img_gen --batch 12 --fp8
  : > $O/bench_lines.txt; RC[bench_lines]=0
  for f in $O/bench_*.json; do
    n=$(basename $f .json)
    if [ "${RC[$n]}" = "0" ] && [ -s $f ]; then echo "# $n" >> $O/bench_lines.txt; cat $f >> $O/bench_lines.txt; else echo "# $n FAILED rc=${RC[$n]}" >> $O/bench_lines.txt; RC[bench_lines]=1; fi
  done
  python - > $O/summary.txt 2>&1 <<'EOF'
import json, glob
for f in sorted(glob.glob("gpurun_out/final/bench_*.json")):
    try:
        j = json.load(open(f))
        print(f, j["value"], j["ms_per_step"], (j.get("roofline") or {}).get("frac"), "packed leg:", (j.get("packed_token_order") or {}).get("value"),
              "parity:", {k: (j.get("parity") or {}).get(k) for k in ("loss_rel", "logits_rel_l2", "storage_model_ratio", "argmax_rate", "argmax_sure_equal")} if j.get("parity") else None,
              "other:", {k: v.get("value") for k, v in (j.get("other_shapes") or {}).items() if isinstance(v, dict)})
    except Exception as e:
        print(f, "UNREADABLE", e)
EOF
  cp $O/bench_lines.txt profiles/${R}_bench_lines_final.txt; mkdir -p $O/profiles; cp $O/bench_lines.txt $O/profiles/${R}_bench_lines_final.txt; echo "keep: ${R}_bench_lines_final.txt (failed legs are marked inside)" >> $O/rc.txt
fi

if has prof; then
  # 3. kernel stats of the default bench command: whole process (--stats) and the timed steps only (markers)
  leg prof 900 $O/prof.log rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-packed-leg --no-shape-legs --no-cfg5-leg
  leg window 300 $O/window.txt python tools/trace_window.py "$(find $O/stats -name '*kernel_trace.csv' | head -1)" 6 $O/${R}_bench_b64_timed_steps.csv
  keep window $O/${R}_bench_b64_timed_steps.csv ${R}_bench_b64_timed_steps.csv
  keep prof "$(find $O/stats -name '*kernel_stats.csv' | head -1)" ${R}_bench_b64_kernel_stats.csv
  leg prof_b3 900 $O/prof_b3.log rocprofv3 --kernel-trace --stats -d $O/stats_b3 -o st --output-format csv -- python3 bench.py --steps 12 --warmup 4 --batch 3 --grad-accum 2 --no-cpu-baseline --no-roofline
  leg window_b3 300 $O/window_b3.txt python tools/trace_window.py "$(find $O/stats_b3 -name '*kernel_trace.csv' | head -1)" 12 $O/${R}_bench_b3ga2_timed_steps.csv
  keep window_b3 $O/${R}_bench_b3ga2_timed_steps.csv ${R}_bench_b3ga2_timed_steps.csv
fi

if has micro; then
  # 3b. micro-benchmarks: HBM-bound kernels, decode, the GEMM family (forms, power, bits)
  leg stamps7 900 $O/stamps7.log bash -c 'for shp in "32768 2560 10240 plain 1" "32768 10240 2560 gelu2 1" "32768 2560 2560 res 1" "131584 1024 4096 res 1"; do python tools/stamp_gemm7.py $shp; set -- $shp; python tools/stamp_gemm3.py $1 $2 $3 pp256a $4 $5; done'
  keep stamps7 $O/stamps7.log ${R}_gemm7_stamps_final.txt
  leg ln 300 $O/bench_ln.log python tools/bench_ln.py --rotate 3
  leg adamw 300 $O/bench_adamw.log python tools/bench_adamw.py
  cat $O/bench_ln.log $O/bench_adamw.log > $O/hbm_kernels.txt; RC[hbm]=$(( ${RC[ln]} + ${RC[adamw]} ))
  keep hbm $O/hbm_kernels.txt ${R}_hbm_kernels_microbench.txt
  leg decode 900 $O/bench_decode.log python tools/bench_decode.py
  keep decode $O/bench_decode.log ${R}_decode_timings.txt
  leg gemm_ab 1800 $O/gemm_ab.log python tools/bench_gemm_ab.py 5 pp256a,pp256b,pp256x,pp256px,w4x,w4x_pf,w8
  keep gemm_ab $O/gemm_ab.log ${R}_gemm_ab_forms.txt
  leg ks_bits 600 $O/ks_bits.log python tools/check_ks_bits.py
  keep ks_bits $O/ks_bits.log ${R}_kstrided_weight_bit_identity.txt
  leg gemm_power 300 $O/gemm_power.log python tools/bench_gemm_power.py
  keep gemm_power $O/gemm_power.log ${R}_gemm_power_zeros_vs_random.txt
  leg variant_bits 600 $O/variant_bits.log python tools/check_variant_bits.py
  keep variant_bits $O/variant_bits.log ${R}_gemm_variant_bit_identity.txt
  leg vendor 600 $O/vendor.log python tools/bench_vendor_gemm.py
  keep vendor $O/vendor.log ${R}_vendor_gemm.txt
  leg mx 300 $O/mx.log python tools/bench_mx.py
  keep mx $O/mx.log ${R}_mx_gemm.txt
  leg mx_step 600 $O/mx_step.log python tools/mx_step_shapes.py
  keep mx_step $O/mx_step.log ${R}_mx_step_shapes_final.txt
  leg attn 600 $O/attn.log python tools/bench_attn2.py
  keep attn $O/attn.log ${R}_attention_microbench.txt
  leg loader 900 $O/loader.log python tools/bench_loader.py 8 8
  keep loader $O/loader.log ${R}_loader_throughput.txt
fi

if has pmc; then
  # 4. PMC passes (each counter group in its own run, --kernel-trace only): the step's dominant GEMM instances and the attention kernels
  export PMC_MANIFEST=$PWD/$O/pmc_manifest.json
  RC[pmcg]=0
  for pm in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $pm | cut -d' ' -f1)
    leg pmcg_$tag 600 $O/pmcg_$tag.log rocprofv3 --kernel-trace --pmc $pm -d $O/pmcg_$tag -o pmc --output-format csv -- python3 tools/pmc_gemm_step.py
    [ "${RC[pmcg_$tag]}" = "0" ] || RC[pmcg]=1
  done
  leg pmc_json 300 $O/pmc_rows.json python tools/pmc_to_json.py $O/pmc_manifest.json $O/pmcg_FETCH_SIZE $O/pmcg_WRITE_SIZE $O/pmcg_SQ_VALU_MFMA_BUSY_CYCLES $O/${R}_pmc_gemm
  [ "${RC[pmcg]}" = "0" ] || RC[pmc_json]=1
  keep pmc_json $O/${R}_pmc_gemm.csv ${R}_pmc_gemm.csv
  keep pmc_json $O/${R}_pmc_gemm.json ${R}_pmc_gemm.json
  RC[pmca]=0
  for pm in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    tag=$(echo $pm | cut -d' ' -f1)
    leg pmca_$tag 600 $O/pmca_$tag.log rocprofv3 --kernel-trace --pmc $pm -d $O/pmca_$tag -o pmc --output-format csv -- python3 tools/pmc_attn.py
    [ "${RC[pmca_$tag]}" = "0" ] || RC[pmca]=1
  done
  leg pmca_sum 300 $O/${R}_pmc_attention.csv python tools/pmc_summary.py $O/pmca_SQ_VALU_MFMA_BUSY_CYCLES $O/pmca_SQ_LDS_BANK_CONFLICT $O/pmca_SQ_WAVE_CYCLES --match attn
  [ "${RC[pmca]}" = "0" ] || RC[pmca_sum]=1
  keep pmca_sum $O/${R}_pmc_attention.csv ${R}_pmc_attention.csv
fi
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete

if has pytest; then
  # 5. the whole GPU suite on the final tree
  leg pytest 3000 $O/pytest.log python -m pytest tests -m gpu -q -rf
  grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -10
fi
echo done >> $O/rc.txt
cat $O/rc.txt; [ -f $O/summary.txt ] && cat $O/summary.txt; [ -f $O/window.txt ] && cat $O/window.txt
