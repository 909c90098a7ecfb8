"""scratch: bf16 / nudged bf16 / fp8 loss curves at cfg5's width (the data behind test_cfg5_width_fp8_loss_curve_against_the_chaos_floor)"""
import sys, os, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import _parity as P
from unimp_amd import functional as F_
from unimp_amd.train import Trainer
lr = float(sys.argv[1]) if len(sys.argv) > 1 else 2e-4
cfg = P.CFG5_SLIM
om, layout = P.build_oracle(cfg)
batches = [{k: v.cuda() for k, v in P.make_batch(cfg, layout, seed=950 + i).items()} for i in range(8)]
for name, fp8, nudge in (("A bf16", False, False), ("B bf16 nudged", False, True), ("C fp8", True, False)):
    F_.FP8_FROZEN = fp8
    hm = P.build_hip(cfg, om, layout)
    if nudge:
        w = next(p for n, p in hm.named_parameters() if p.requires_grad and p.dim() == 2)
        with torch.no_grad():
            v = w.view(-1)[:1].view(torch.int16); v += 1
    tr = Trainer(hm, layout.special(), lr=lr, lr_scheduler="constant", gamma=2.0)
    c = [tr.step(batches[i % 8])[0].item() for i in range(32)]
    print(name, " ".join(f"{x:.3f}" for x in c), flush=True)
    tr.dp.remove(); del tr, hm; torch.cuda.empty_cache()
