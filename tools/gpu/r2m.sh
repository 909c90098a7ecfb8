cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2m; mkdir -p $O
for g in 512 1024 2048 4096 8192 100000; do echo "== grid cap $g" >> $O/ln.log; UNIMP_LN_GRID=$g timeout 200 python tools/bench_ln.py 2>&1 | grep -v amdgpu >> $O/ln.log; done
