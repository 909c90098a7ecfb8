#!/bin/bash
# ViT attention forward: last key as softmax seed + waves per block, A/B
mkdir -p gpurun_out/r4r; O=gpurun_out/r4r
for f in 0 5 3 9; do UNIMP_ATTN_VIT=$f timeout 300 python tools/scratch/vit_attn_ab.py >> $O/ab.txt 2>&1; done
cat $O/ab.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -3 | tee $O/pytest.txt
