#!/bin/bash
# round 6, first pass: the default bench line on this round's tree (gradient parity object, explain-template batches) + the tests the ABI-8 / ADVICE changes touch
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
python - <<'PY' > $O/summary.txt 2>&1
import json
d = json.load(open("gpurun_out/r06_a/bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
p = d.get("parity", {})
print({k: p.get(k) for k in ("loss_rel_mean", "loss_rel_max", "loss_rel_per_batch", "loss_rel_pooled", "storage_model_ratio", "argmax_rate", "oracle_seconds", "labeled_positions", "error")})
print(json.dumps(p.get("gradients"), indent=1))
print(p.get("vit_257th_key_ab"))
print({k: (v.get("value"), v.get("roofline_frac")) for k, v in d.get("other_shapes", {}).items() if isinstance(v, dict)})
print(d.get("cfg5_imggen_fp8", {}).get("value"), d.get("cfg5_fp8", {}).get("value"), d.get("cfg4_hm", {}).get("value"))
PY
cat $O/summary.txt
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "attention3 or attn" > $O/pytest_attn.log 2>&1; echo "pytest attn rc=$?" >> $O/rc.txt; tail -3 $O/pytest_attn.log
timeout 1200 python -m pytest tests/test_dp_gpu.py tests/test_model_gpu.py -q -x -m gpu > $O/pytest_dp_model.log 2>&1; echo "pytest dp/model rc=$?" >> $O/rc.txt; tail -3 $O/pytest_dp_model.log
cat $O/rc.txt
