"""What the 8-bit stored act'(z) (functional.DERIV_U8, VERDICT r2 #2b) does to the gradients -- a script, not a collected test:

    python tests/deriv_u8_gradients.py [CFG2_SLIM]

One backward pass of the HIP model at cfg2 width with the derivative stored as bf16 and as uint8, both against the fp32 oracle's
gradients on the same weights and batch: per trainable tensor rel-L2, worst and median.  Test infrastructure: imports oracle/."""
import os
import sys
import statistics
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _parity as P                                   # noqa: E402
from unimp_amd import functional as F_                # noqa: E402
from unimp_amd.train import Trainer                   # noqa: E402

cfgname = sys.argv[1] if len(sys.argv) > 1 else "CFG2_SLIM"
cfg = getattr(P, cfgname)
om, layout = P.build_oracle(cfg)
batch = P.make_batch(cfg, layout, seed=31)
_, loss_o, _, og = P.oracle_step(om, layout, batch)
og = {n: g.float() for n, g in og.items()}
res = {}
for u8 in (False, True):
    F_.DERIV_U8 = u8
    hm = P.build_hip(cfg, om, layout)
    tr = Trainer(hm, layout.special(), lr=1e-3, gamma=2.0)
    hm.train()
    loss, stats, out, labels = tr.forward_loss({k: v.cuda() for k, v in batch.items()})
    tr._backward(loss)
    res[u8] = ({n: p.grad.float().cpu().clone() for n, p in hm.named_parameters() if p.grad is not None}, float(loss))
    tr.dp.remove()
print(f"[{cfgname}] loss: oracle {float(loss_o):.6f}, bf16 act' {res[False][1]:.6f}, uint8 act' {res[True][1]:.6f}")
rows = []
for n, g in og.items():
    if n not in res[False][0] or float(g.abs().max()) == 0:
        continue
    rows.append((n, P.rel_l2(res[False][0][n], g), P.rel_l2(res[True][0][n], g), P.rel_l2(res[True][0][n], res[False][0][n])))
for label, i in (("bf16 act' vs oracle", 1), ("uint8 act' vs oracle", 2), ("uint8 vs bf16 act'", 3)):
    v = [r[i] for r in rows]
    w = max(rows, key=lambda r: r[i])
    print(f"  {label:22s}: {len(v)} tensors, median rel-L2 {statistics.median(v):.3e}, worst {w[i]:.3e} ({w[0]})")
