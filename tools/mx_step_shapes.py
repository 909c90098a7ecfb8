"""Per-shape table of the MX-fp8 GEMM launches inside the cfg5-family train step (9b Flamingo, frozen towers in MX-fp8): ms per step, TFLOP/s, share."""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from unimp_amd import ops, functional as F_
from unimp_amd.synthetic import make_batch
from unimp_amd.train import Trainer
dev = torch.device("cuda")
F_.FP8_FROZEN = True
m9, lay9 = bench.build_cfg2(dev, lang="anas-awadalla/mpt-7b", every=4)
tr = Trainer(m9, lay9.special(), lr=2e-4, weight_decay=0.1, gamma=2.0, use_reweight=True, lr_scheduler="cosine", warmup_steps=10, total_steps=10000)
pool = [make_batch(lay9, 24, 8, 512, seed=8642 + 1000 * i, device=dev, vision_dtype=torch.bfloat16) for i in range(6)]
for i in range(3):
    tr.step(pool[i])
torch.cuda.synchronize()
ops.GEMM_PROFILE = []
N = 3
for i in range(N):
    tr.step(pool[3 + i])
torch.cuda.synchronize()
pr, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for e0, e1, fl, key in pr:
    if key[-1] != "mxfp8":
        continue
    a = agg[key[:4]]
    a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += fl
tot = sum(a[1] for a in agg.values())
for k, (c, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"mx M={k[0]:6d} N={k[1]:6d} K={k[2]:6d} {k[3]:18s} calls/step {c // N:4d}  {ms / N:8.2f} ms/step  {fl / ms / 1e9:7.0f} TFLOP/s  {100 * ms / tot:5.1f} %")
print(f"total {tot / N:.2f} ms/step, {sum(a[2] for a in agg.values()) / tot / 1e9:.0f} TFLOP/s")
