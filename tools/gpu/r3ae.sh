cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ae; mkdir -p $O
timeout 900 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -k "packed" 2>&1 | tail -3
