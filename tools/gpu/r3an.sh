cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=r03
O=gpurun_out/r3an; mkdir -p $O
export PMC_MANIFEST=$PWD/$O/pmc_manifest.json
for pm in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $pm | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pm -d $O/pmcg_$tag -o pmc --output-format csv -- python3 tools/pmc_gemm_step.py > $O/pmcg_$tag.log 2>&1
done
python tools/pmc_to_json.py $O/pmc_manifest.json $O/pmcg_FETCH_SIZE $O/pmcg_WRITE_SIZE $O/pmcg_SQ_VALU_MFMA_BUSY_CYCLES $O/${R}_pmc_gemm > $O/pmc_rows.json 2> $O/pmc_to_json.err
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
cat $O/pmc_to_json.err | head -5
cut -d, -f1,9,10,15 $O/${R}_pmc_gemm.csv | tail -11
