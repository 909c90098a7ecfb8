"""training-step soak at cfg2: memory must be flat from step to step and the loss finite."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from unimp_amd.synthetic import make_batch
from unimp_amd.train import Trainer
dev = torch.device("cuda")
model, layout = bench.build_cfg2(dev)
tr = Trainer(model, layout.special(), lr=2e-4, total_steps=1000)
for i in range(24):
    b = make_batch(layout, 16, 8, 512, seed=i, device=dev, vision_dtype=torch.bfloat16)       # a new batch every step
    loss, stats = tr.step(b)
    if i % 4 == 3:
        torch.cuda.synchronize()
        print(f"step {i + 1:3d}: loss {float(loss):8.4f}  allocated {torch.cuda.memory_allocated() / 2**30:7.2f} GiB  reserved {torch.cuda.memory_reserved() / 2**30:7.2f} GiB  "
              f"peak {torch.cuda.max_memory_allocated() / 2**30:7.2f} GiB")
