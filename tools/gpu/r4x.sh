#!/bin/bash
# loader beside the step after the deferred labeled-row count (no blocking sync at the top of the step)
O=gpurun_out/r4x; mkdir -p $O
timeout 900 python tools/bench_loader.py 8 8 500 2>&1 | grep -v amdgpu | tee $O/loader.txt
UNIMP_ROWS_SYNC=1 timeout 900 python tools/bench_loader.py 8 8 500 2>&1 | grep -v amdgpu | tee $O/loader_rows_sync.txt
