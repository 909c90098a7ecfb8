import sys, os, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import _parity as P
from unimp_amd import functional as F_
from unimp_amd.train import Trainer
cfg = P.TINY_MX
om, layout = P.build_oracle(cfg)
batches = [{k: v.cuda() for k, v in P.make_batch(cfg, layout, seed=900 + i).items()} for i in range(8)]
for flag in (False, True, True):
    F_.FP8_FROZEN = flag; F_.FP8_MIN_DIM = 128
    hm = P.build_hip(cfg, om, layout)
    tr = Trainer(hm, layout.special(), lr=2e-3, lr_scheduler="constant", gamma=2.0)
    c = [tr.step(batches[i % 8])[0].item() for i in range(60)]
    print("fp8" if flag else "bf16", " ".join(f"{x:.2f}" for x in c), flush=True)
    tr.dp.remove()
