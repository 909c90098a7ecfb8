#!/bin/bash
# full GPU suite on the final tree + one more default bench line (another box of the pool)
O=gpurun_out/r4y; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt; tail -2 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
python -c "import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cfg5_fp8']['mx_gemms']['frac'], d['other_shapes']['reference_shape_b3_ga2']['value'])"
