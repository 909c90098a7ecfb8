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
cases = [("eval_rec   K=10,  50 new", 10, 50, 1), ("eval_rec   K=10,  50 new, 4 users per call", 10, 50, 4), ("eval_exp   K=5,  256 new", 5, 256, 1),
         ("eval_img_gen greedy, 600 new", 1, 600, 1)]
if quick:
    cases = cases[:2]
modes = [("split-key kernel, prompt keys read once per prompt", True, True), ("split-key kernel, every beam reads its own copy", True, False),
         ("training kernel on one query row", False, False)]
for name, K, new, users in cases:
    for label, dec, shared in modes:
        if K == 1 and dec and not shared:
            continue                       # no beams: the two split-key modes are the same launch
        ops.DECODE_ATTN, ops.DECODE_SHARED_PREFIX = dec, shared
        kw = dict(num_beams=K, num_return_sequences=K, early_stopping=False, max_new_tokens=new, eos_token_id=-1, pad_token_id=layout.eos)
        ii, vv = ids.repeat(users, 1), vx.repeat(users, 1, 1, 1, 1, 1)
        model.generate(vv, ii, **{**kw, "max_new_tokens": 3})
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model.generate(vv, ii, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / users
        print(f"{name:44s} {label:52s}: prompt {ids.shape[1]} tokens, {out.shape[1] - ids.shape[1]} new tokens: {dt:.3f} s per user "
              f"({dt * users / new * 1e3:.2f} ms per token-step incl. vision + prefill)", flush=True)
ops.DECODE_ATTN = ops.DECODE_SHARED_PREFIX = True
