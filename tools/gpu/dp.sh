cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/dp; mkdir -p $O
timeout 900 python -m pytest tests/test_dp_gpu.py tests/test_model_gpu.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 6 --warmup 2 --cpu-full-steps 0 > $O/bench_tr.json 2> $O/bench_tr.err; echo "torchrun bench rc=$?" >> $O/rc.txt
