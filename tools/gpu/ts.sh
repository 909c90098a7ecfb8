cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/ts; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "gemm" -x > $O/kern.log 2>&1; echo "kern rc=$?" > $O/rc.txt
for v in 0 1 0 1; do
UNIMP_GEMM_TAIL_SPLIT=$v timeout 900 python bench.py --steps 10 --warmup 3 --cpu-full-steps 0 > $O/bench_$v.json 2> $O/bench_$v.err; echo "ts$v rc=$?" >> $O/rc.txt
python -c "
import json;d=json.load(open('$O/bench_$v.json'));print('ts$v',d['value'],d['ms_per_step'],d['roofline']['gemm_ms_per_step'],d['config']['loss'],d['config']['gemm_autotune'])" >> $O/sum.txt
done
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -x > $O/model.log 2>&1; echo "model rc=$?" >> $O/rc.txt
