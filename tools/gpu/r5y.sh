#!/bin/bash
# bench lines of the final tree once more, with the side legs timed outside their GEMM-profiling steps (bench.py of the last commit)
ROUND=r05 bash tools/gpu/final.sh bench
tail -16 gpurun_out/final/rc.txt
