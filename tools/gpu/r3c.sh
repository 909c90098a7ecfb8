# round 3, third GPU pass: full suite; b=3 GA=2 eager vs graph; kernel trace window of the default bench; PMC passes with the new tools
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -rf -s > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -30
timeout 600 python bench.py --no-cpu-baseline --batch 3 --grad-accum 2 > $O/bench_b3ga2.json 2> $O/bench_b3ga2.err; echo "b3 rc=$?" >> $O/rc.txt
timeout 600 python bench.py --no-cpu-baseline --batch 3 --grad-accum 2 --graph > $O/bench_b3ga2_graph.json 2> $O/bench_b3ga2_graph.err; echo "b3graph rc=$?" >> $O/rc.txt
timeout 600 python bench.py --no-cpu-baseline --batch 3 --grad-accum 2 --dense-head-backward > $O/bench_b3ga2_dense.json 2> $O/bench_b3ga2_dense.err; echo "b3dense rc=$?" >> $O/rc.txt
timeout 600 python bench.py --no-cpu-baseline --batch 16 --graph > $O/bench_b16_graph.json 2> $O/bench_b16_graph.err
timeout 600 python bench.py --no-cpu-baseline --batch 16 > $O/bench_b16.json 2> $O/bench_b16.err
for f in $O/bench_b*.json; do python -c "import json,sys; j=json.load(open('$f')); print('$f', j['value'], j['ms_per_step'])"; done
# kernel trace of the default bench, cut to the timed steps
timeout 900 rocprofv3 --kernel-trace -d $O/trace -o tr --output-format csv -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace.log 2>&1
python tools/trace_window.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 6 $O/r03_bench_b64_timed_steps.csv
timeout 900 rocprofv3 --kernel-trace -d $O/trace3 -o tr --output-format csv -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --batch 3 --grad-accum 2 > $O/trace3.log 2>&1
python tools/trace_window.py $(find $O/trace3 -name "*kernel_trace.csv" | head -1) 6 $O/r03_bench_b3ga2_timed_steps.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
# PMC passes over the pinned GEMM instances
export PMC_MANIFEST=$PWD/$O/pmc_manifest.json
for pm in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $pm | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pm -d $O/pmcg_$tag -o pmc --output-format csv -- python3 tools/pmc_gemm_step.py > $O/pmcg_$tag.log 2>&1
done
python tools/pmc_to_json.py $O/pmc_manifest.json $O/pmcg_FETCH_SIZE $O/pmcg_WRITE_SIZE $O/pmcg_SQ_VALU_MFMA_BUSY_CYCLES $O/r03_pmc_gemm > $O/pmc_rows.json 2> $O/pmc_to_json.err; echo "pmc rc=$?" >> $O/rc.txt
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
cat $O/rc.txt
