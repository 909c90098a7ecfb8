cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2i; mkdir -p $O
export ATTN=lm REP=3
for d in 7 15 31 63; do UNIMP_A2_DBG=$d timeout 300 rocprofv3 --kernel-trace --stats -d $O/st$d -o st --output-format csv -- python3 tools/pmc_attn.py > $O/st$d.log 2>&1; done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
