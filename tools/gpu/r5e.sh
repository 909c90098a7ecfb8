# round 5: end-of-round measurement, first half (fresh autotune table, bench lines, kernel traces)
ROUND=r05 bash tools/gpu/final.sh tune bench prof
