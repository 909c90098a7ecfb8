"""host time of one optimizer step at the reference's shape (b = 3 x GA 2, fused) vs its device time: is the step launch-bound?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from unimp_amd.synthetic import make_batch
from unimp_amd.train import Trainer
dev = torch.device("cuda")
model, layout = bench.build_cfg2(dev)
tr = Trainer(model, layout.special(), lr=2e-4, weight_decay=0.1, gamma=2.0, use_reweight=True, lr_scheduler="cosine", warmup_steps=10, total_steps=10000, grad_accum=2)
pool = [make_batch(layout, 3, 8, 512, seed=1234 + 1000 * i, device=dev, vision_dtype=torch.bfloat16) for i in range(8)]
for i in range(8):
    tr.step(pool[i])
torch.cuda.synchronize()
N = 20
host = []
t0 = time.perf_counter()
for s in range(N):
    a = time.perf_counter()
    tr.step(pool[(2 * s) % 8]); tr.step(pool[(2 * s + 1) % 8])
    host.append(time.perf_counter() - a)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N
host.sort()
print(f"b3 x GA2: wall {wall * 1e3:.2f} ms per optimizer step; host time inside step(): median {host[N // 2] * 1e3:.2f} ms, min {host[0] * 1e3:.2f}, max {host[-1] * 1e3:.2f}")
