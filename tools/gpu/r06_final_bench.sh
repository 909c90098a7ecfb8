#!/bin/bash
ROUND=r06 bash tools/gpu/final.sh bench prof
cat gpurun_out/final/rc.txt | tail -40; cat gpurun_out/final/summary.txt | cut -c1-400
