"""ad-hoc: 1024 tiles whose operand rows all alias ONE row (row stride 0): full-chip MFMA power, no memory-system load."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
torch.manual_seed(0)
K = 8192
for mode in ("real", "alias-rows", "alias-tiles"):
    m = n = 8192
    if mode == "real":
        a = torch.randn(m, K, device="cuda").to(torch.bfloat16); b = torch.randn(n, K, device="cuda").to(torch.bfloat16)
    elif mode == "alias-rows":
        a = torch.randn(1, K, device="cuda").to(torch.bfloat16).expand(m, K); b = torch.randn(1, K, device="cuda").to(torch.bfloat16).expand(n, K)
    else:   # every tile reads the same 256 x K panels: as_strided with a wrap every 256 rows is not expressible -> use 256-row operands, 1 tile
        continue
    out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    for variant in ("pp256", "w8", "w4"):
        try:
            for _ in range(3):
                ops.gemm(a, b, out=out, variant=variant)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm(a, b, out=out, variant=variant)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            print(f"{mode:12s} {variant}: {us:7.1f} us  {2.0 * m * n * K / us / 1e6:7.1f} TF", flush=True)
        except Exception as e:
            print(mode, variant, "failed:", str(e)[:200])
