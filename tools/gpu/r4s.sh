#!/bin/bash
# ViT attention seed form: model-level tests + default bench line
O=gpurun_out/r4s; mkdir -p $O
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_widths_gpu.py tests/test_preprocess_gpu.py -m gpu -q -x > $O/pytest_model.log 2>&1; echo "pytest model rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_model.log | tail -5
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open("gpurun_out/r4s/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d.get("parity"))
P
