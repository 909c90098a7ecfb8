"""GEMM micro-benchmark on the step's real shapes (random data): TFLOP/s per operand form, v1 (128^2) vs v2 (256-row).
usage: python tools/bench_gemm.py [tokens]      env UNIMP_GEMM_V1=1 forces the 128x128 kernel"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
shapes = [("fwd qkv", M, 7680, 2560, 0, 0), ("fwd dense", M, 2560, 2560, 0, 0), ("fwd up", M, 10240, 2560, 0, 0),
          ("fwd down", M, 2560, 10240, 0, 0), ("dX up (N=2560,K=10240)", M, 2560, 10240, 0, 1), ("dX down", M, 10240, 2560, 0, 1),
          ("dX qkv", M, 2560, 7680, 0, 1), ("dW ff1 [10240,2560]", 10240, 2560, M, 1, 1), ("dW ff2 [2560,10240]", 2560, 10240, M, 1, 1),
          ("xattn q", M, 512, 2560, 0, 0), ("vit fc1", 8 * 257 * (M // 512), 4096, 1024, 0, 0), ("head", M, 74053, 2560, 0, 0)]
torch.manual_seed(0)
for name, m, n, k, aks, bks in shapes:
    a = torch.randn((k, m) if aks else (m, k), device="cuda").to(torch.bfloat16)
    b = torch.randn((k, n) if bks else (n, k), device="cuda").to(torch.bfloat16)
    ldc = (n + 7) // 8 * 8
    out = torch.empty((m, ldc), dtype=torch.bfloat16, device="cuda")[:, :n]
    res = []
    for variant in ("v1", "dma256", "dma128", "pp256", "pp128"):
        for _ in range(2):
            ops.gemm(a, b, a_ks=bool(aks), b_ks=bool(bks), out=out, variant=variant)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 10
        e0.record()
        for _ in range(it):
            ops.gemm(a, b, a_ks=bool(aks), b_ks=bool(bks), out=out, variant=variant)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / it
        res.append(f"{variant} {2*m*n*k/ms/1e9:7.1f}")
    print(f"{name:24s} M={m:6d} N={n:6d} K={k:6d}  TFLOP/s: " + " | ".join(res), flush=True)
