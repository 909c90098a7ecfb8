cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ac; mkdir -p $O
for f in 0.75 0.5 1.0; do timeout 300 python tools/bench_attn_packed.py $f 2>&1 | grep -v amdgpu; done | tee $O/attn_packed.log
