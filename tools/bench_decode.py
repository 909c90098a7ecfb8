"""Decode-loop timing at cfg2 size (eval_rec.py:100-110: K = 10 beams, 50 new tokens).  Not a pytest file.
usage: python tools/bench_decode.py [new_tokens] [beams]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from unimp_amd.synthetic import make_batch

new = int(sys.argv[1]) if len(sys.argv) > 1 else 50
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda")
model, layout = bench.build_cfg2(dev, gate=0.5)
bt = make_batch(layout, 1, 8, 512, seed=7, device="cuda", vision_dtype=torch.bfloat16)
n = int(bt["attention_mask"][0].sum())
ids, vx = bt["lang_x"][:1, :n - 2], bt["vision_x"][:1]
kw = dict(num_beams=K, num_return_sequences=K, early_stopping=False, max_new_tokens=new, eos_token_id=-1, pad_token_id=layout.eos)
modes = [(True, True), (True, False)] + ([(False, False)] if os.environ.get("RESCORE") else [])
for uc, ug in modes:
    model.generate(vx, ids, use_cache=uc, use_graph=ug, **{**kw, "max_new_tokens": 3})
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = model.generate(vx, ids, use_cache=uc, use_graph=ug, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"use_cache={uc} graph={ug}: prompt {ids.shape[1]} tokens, K={K}, {out.shape[1] - ids.shape[1]} new tokens: {dt:.3f} s "
          f"({dt / new * 1e3:.2f} ms/token-step incl. vision+prefill)")
