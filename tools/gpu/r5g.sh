# round 5: end-of-round measurement, second half (micro-benchmarks, PMC passes, the whole GPU suite)
ROUND=r05 bash tools/gpu/final.sh micro pmc pytest
