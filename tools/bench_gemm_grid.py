"""ad-hoc: time of ONE round of 256x256 tiles (K = 8192) as the number of busy CUs grows -- separates the kernel's own schedule from
memory-system contention."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
torch.manual_seed(0)
K = 8192
for (m, n) in ((256, 256), (1024, 1024), (2048, 2048), (4096, 2048), (4096, 4096), (8192, 4096), (8192, 8192)):
    a = torch.randn(m, K, device="cuda").to(torch.bfloat16)
    b = torch.randn(n, K, device="cuda").to(torch.bfloat16)
    out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    res = []
    for variant in ("pp256", "w8", "w4"):
        for _ in range(3):
            ops.gemm(a, b, out=out, variant=variant)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm(a, b, out=out, variant=variant)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        tiles = (m // 256) * (n // 256)
        rounds = -(-tiles // 256)
        res.append(f"{variant} {us:7.1f} us  {us / rounds:6.1f} us/round  per-CU {2.0 * 256 * 256 * K / (us / rounds) / 1e6:5.2f} TF")
    print(f"tiles {tiles:5d}: " + " | ".join(res), flush=True)
