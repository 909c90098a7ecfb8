import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from unimp_amd.train import Trainer
from unimp_amd import ops
cfg = P.TINY
om, layout = P.build_oracle(cfg)
bad = 0
names = None
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    batch = {k: v.cuda() for k, v in P.make_batch(cfg, layout, seed=500 + it).items()}
    outs = []
    for rep in range(2):
        tr = Trainer(P.build_hip(cfg, om, layout), layout.special(), lr=1e-3, lr_scheduler="constant", bucket_bytes=1 << 16)
        l, _ = tr.step(batch)
        torch.cuda.synchronize()
        outs.append((l.item(), tr.opt.flat_p.clone(), tr.opt.flat_g.clone() if hasattr(tr.opt, "flat_g") else None))
    same_l = outs[0][0] == outs[1][0]
    same_p = torch.equal(outs[0][1], outs[1][1])
    if not (same_l and same_p):
        bad += 1
        print(f"iter {it}: loss equal {same_l} ({outs[0][0]!r} vs {outs[1][0]!r}), params equal {same_p}, differing params {(outs[0][1] != outs[1][1]).sum().item()}")
print("nondeterministic iterations:", bad)
