#!/bin/bash
# final measurement pass 1 of round 5 on the final tree (attention3 in): bench lines + kernel traces
ROUND=r05 bash tools/gpu/final.sh bench prof
cat gpurun_out/final/rc.txt | tail -30
