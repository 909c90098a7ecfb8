"""Decode-loop timing at cfg2 size for the reference's eval calls: eval_rec.py:100-110 (K = 10 beams, 50 new tokens), eval_exp.py:
103-113 (5 beams, 256 new tokens) and eval_img_gen.py:102-111 (greedy, 600 new tokens), each with the split-key decode attention
kernel (csrc/decode_attn.hip, default) and with the training kernel on one query row (UNIMP_DECODE_ATTN=0's path).  Not a pytest file.
usage: python tools/bench_decode.py [quick]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch           # noqa: E402
import bench           # noqa: E402
from unimp_amd import ops        # noqa: E402
from unimp_amd.synthetic import make_batch       # noqa: E402

quick = len(sys.argv) > 1
dev = torch.device("cuda")
model, layout = bench.build_cfg2(dev, gate=0.5)
bt = make_batch(layout, 1, 8, 512, seed=7, device="cuda", vision_dtype=torch.bfloat16)
n = int(bt["attention_mask"][0].sum())
ids, vx = bt["lang_x"][:1, :n - 2], bt["vision_x"][:1]
cases = [("eval_rec   K=10,  50 new", 10, 50), ("eval_exp   K=5,  256 new", 5, 256), ("eval_img_gen greedy, 600 new", 1, 600)]
if quick:
    cases = cases[:1]
for name, K, new in cases:
    for dec in (True, False):
        ops.DECODE_ATTN = dec
        kw = dict(num_beams=K, num_return_sequences=K, early_stopping=False, max_new_tokens=new, eos_token_id=-1, pad_token_id=layout.eos)
        model.generate(vx, ids, **{**kw, "max_new_tokens": 3})
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model.generate(vx, ids, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name:30s} decode attention = {'split-key kernel' if dec else 'training kernel, 1 query row'}: prompt {ids.shape[1]} tokens, "
              f"{out.shape[1] - ids.shape[1]} new tokens: {dt:.3f} s per user ({dt / new * 1e3:.2f} ms per token-step incl. vision + prefill)", flush=True)
ops.DECODE_ATTN = True
