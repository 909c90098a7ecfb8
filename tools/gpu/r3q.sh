# A/B of the LDS tile pitch of the first-generation attention kernels (dK/dV in production): rebuild attention.o with the old pitch, bench, rebuild
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q; mkdir -p $O
echo "== new pitch (32 mod 64)" > $O/attn_ab.log
timeout 600 python tools/bench_attn2.py lm vit xattn perc mpt >> $O/attn_ab.log 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "attention" > $O/pytest_attn.log 2>&1; tail -1 $O/pytest_attn.log
cp unimp_amd/libunimp_hip.so $O/lib_new.so
touch unimp_amd/csrc/attention.hip; make -C unimp_amd/csrc EXTRA=-DATTN_PITCH_R2 > $O/make_old.log 2>&1
echo "== old pitch (+16)" >> $O/attn_ab.log
timeout 600 python tools/bench_attn2.py lm vit xattn perc mpt >> $O/attn_ab.log 2>&1
cp $O/lib_new.so unimp_amd/libunimp_hip.so; rm $O/lib_new.so
grep -v amdgpu.ids $O/attn_ab.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench_new.json 2> $O/bench_new.err
python -c "import json; j=json.load(open('$O/bench_new.json')); print('bench new pitch', j['value'], j['ms_per_step'])"
