"""Decode-row GEMM microbench: achieved weight bandwidth of the skinny kernel per cfg2 decode shape.  Not a pytest file."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimp_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 10
shapes = [(7680, 2560), (2560, 2560), (10240, 2560), (2560, 10240), (512, 2560), (2560, 512), (74053, 2560)]
tot_b, tot_t = 0, 0
for N, K in shapes:
    a = torch.randn(M, K, device="cuda").bfloat16()
    ws = [torch.randn(N, K, device="cuda").bfloat16() for _ in range(max(2, int(1.2e9 / (N * K * 2))))]   # rotate: defeat the 256 MB MALL
    ldc = (N + 7) // 8 * 8
    g_ = torch.ones(K, device="cuda").bfloat16(); b_ = torch.zeros(K, device="cuda").bfloat16()
    for variant in ("skinny",) + (("skinny+LN",) if (os.environ.get("SKINNY_LN") and ops.skinny_ln_ok(M, K)) else ()):
        kw = dict(ln=(g_, b_, 1e-5)) if variant.endswith("LN") else dict(variant=variant)
        for w in ws[:2]:
            ops.gemm(a, w, ldc=ldc, **kw)
        g = torch.cuda.CUDAGraph()
        outs = []
        with torch.cuda.graph(g):
            for w in ws:
                outs.append(ops.gemm(a, w, ldc=ldc, **kw))
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (5 * len(ws))
        print(f"M={M} N={N:6d} K={K:6d} {variant:9s} {us:8.1f} us  {N * K * 2 / us / 1e6:7.2f} TB/s")
