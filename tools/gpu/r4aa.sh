#!/bin/bash
# dQ kernel: per-element mask behind a wave-uniform branch -- attention tests + micro-benchmark (compare profiles/r04_attention_microbench.txt)
O=gpurun_out/r4aa; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention or attn" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 600 python tools/bench_attn2.py lm vit xattn mpt lm2k 2>&1 | grep -v amdgpu | tee $O/attn.txt | cut -c1-200
