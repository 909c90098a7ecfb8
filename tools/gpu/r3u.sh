cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3u; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "packed_rows or rope" > $O/k.log 2>&1; tail -15 $O/k.log | cut -c1-300
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q -k "packed" > $O/m.log 2>&1; tail -25 $O/m.log | cut -c1-300
