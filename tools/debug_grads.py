import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")      # run from the repo root
import _parity as P
from unimp_amd.train import Trainer
for name in sys.argv[1:] or ["TINY_OPT"]:
    cfg = getattr(P, name)
    om, layout = P.build_oracle(cfg)
    hm = P.build_hip(cfg, om, layout)
    batch = P.make_batch(cfg, layout)
    wl, wloss, wlab, wg = P.oracle_step(om, layout, batch)
    tr = Trainer(hm, layout.special())
    hm.train()
    loss, stats, out, labels = tr.forward_loss({k: v.cuda() for k, v in batch.items()})
    print(name, "loss", loss.item(), wloss.item(), "logits", P.rel_l2(out["logits"], wl))
    loss.backward()
    named = dict(hm.named_parameters())
    for n, g in wg.items():
        e = P.rel_l2(named[n].grad, g)
        if e > 1.5e-2:
            print(f"  {n:70s} {e:.4f}  |g|={g.norm():.3e}")
