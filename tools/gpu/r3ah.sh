cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ah; mkdir -p $O
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_widths_gpu.py tests/test_fullsize_gpu.py tests/test_dp_gpu.py -m gpu -q -rf > $O/m.log 2>&1; tail -15 $O/m.log | cut -c1-300
