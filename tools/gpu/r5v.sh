#!/bin/bash
# last measurement pass of round 5 on the final tree: bench lines, kernel traces, the GPU suite
ROUND=r05 bash tools/gpu/final.sh bench prof pytest
tail -12 gpurun_out/final/rc.txt; tail -3 gpurun_out/final/pytest.log
