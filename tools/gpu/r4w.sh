#!/bin/bash
# MX-fp8: fixed epilogue kinds + uint8 derivative + LM-only routing: tests, per-shape table of the cfg5 step, A/B of the kinds
O=gpurun_out/r4w; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "mx" > $O/pytest_k.log 2>&1; echo "pytest kernels rc=$?"; tail -3 $O/pytest_k.log
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_widths_gpu.py -m gpu -q -x -k "fp8 or labeled_rows" > $O/pytest_m.log 2>&1; echo "pytest model rc=$?"; grep -E "passed|failed|\[cfg5|\[fp8" $O/pytest_m.log | tail -8
timeout 600 python tools/scratch/mx_step_shapes.py 2>&1 | grep -v amdgpu | tee $O/mx_step_shapes.txt
UNIMP_MX_FIXED_EPI=0 timeout 600 python tools/scratch/mx_step_shapes.py 2>&1 | grep -v amdgpu | tee $O/mx_step_shapes_general.txt | tail -1
