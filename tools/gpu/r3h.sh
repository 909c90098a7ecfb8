# round 3, eighth GPU pass: batch-invariance diagnostic (9b), skinny 16-wave A/B, full-suite durations
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
timeout 900 python tools/debug_batch_invariance.py 9b 8 > $O/invariance_9b.log 2>&1; cat $O/invariance_9b.log | tail -12
for nw in 0 16; do
  echo "UNIMP_SKINNY_NW=$nw (0 = default rule)"; UNIMP_SKINNY_NW=$nw timeout 300 python tools/bench_skinny.py 10 2>&1 | tail -8
  UNIMP_SKINNY_NW=$nw timeout 300 python tools/bench_skinny.py 40 2>&1 | tail -8
done > $O/skinny.log 2>&1; cat $O/skinny.log
UNIMP_SKINNY_AUTO16=1 timeout 600 python tools/bench_decode.py quick > $O/decode_auto16.log 2>&1; grep "every beam" $O/decode_auto16.log
timeout 600 python tools/bench_decode.py quick > $O/decode_default.log 2>&1; grep "every beam" $O/decode_default.log
timeout 2400 python -m pytest tests -m gpu -q -rf --durations=25 > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "^(FAILED|ERROR)|passed|failed|s call|s setup" $O/pytest.log | tail -40
