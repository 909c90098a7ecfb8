import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from unimp_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for N, K in [(10240, 2560), (5120, 5120), (2560, 10240), (20480, 1280), (1280, 20480), (2560, 2560), (5120, 1280), (1280, 5120), (6400, 4096), (3200, 8192)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    ws = [torch.randn(N, K, device="cuda").bfloat16() for _ in range(max(2, int(1.2e9 / (N * K * 2))))]
    for w in ws[:2]:
        ops.gemm(a, w)
    g = torch.cuda.CUDAGraph()
    outs = []
    with torch.cuda.graph(g):
        for w in ws:
            outs.append(ops.gemm(a, w))
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * len(ws))
    print(f"M={M} N={N:6d} K={K:6d} {us:8.1f} us  {N * K * 2 / us / 1e6:7.2f} TB/s  ({N*K*2/1e6:.0f} MB)")
