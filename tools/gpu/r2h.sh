cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2h; mkdir -p $O
for d in 0 1 2 3 4 7; do echo "== dbg $d" >> $O/dbg.log; UNIMP_A2_DBG=$d timeout 300 python tools/bench_attn2.py lm vit 2>&1 | grep -v amdgpu >> $O/dbg.log; done
