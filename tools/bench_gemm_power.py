"""Is the bf16 GEMM rate on this part set by the schedule or by the clock the chip holds under load?  The same kernel, the same
shape, three operand fills: N(0, 1) random bf16 (what bench.py's random-init model feeds the MFMAs), all zeros, and a constant.
Cycles per MFMA do not depend on the data (MI355X_MICROARCH.md, Matrix cores); the toggling power does, and with it the clock
(DVFS give-back, item 1: zero-filled inputs +19 % TF/s at +0.1 % wave cycles).  Prints TFLOP/s per fill and variant; the ratio
zeros / random is the share of the spec peak that the power cap, not the kernel, takes on random data."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops           # noqa: E402

bf = torch.bfloat16
shapes = [("8192^3", 8192, 8192, 8192), ("LM up-proj", 32768, 10240, 2560), ("LM dX up-proj (KC,KS)", 32768, 2560, 10240)]
for name, M, N, K in shapes:
    bks = "KS" in name
    fills = {"randn": lambda *s: torch.randn(*s, device="cuda").to(bf), "zeros": lambda *s: torch.zeros(*s, device="cuda", dtype=bf),
             "const 1.0": lambda *s: torch.ones(*s, device="cuda", dtype=bf)}
    for fname, mk in fills.items():
        a, b = mk(M, K), mk(K, N) if bks else mk(N, K)
        out = torch.empty((M, N), dtype=bf, device="cuda")
        res = []
        for variant in ("pp256", "pp256p", "w8"):
            f = lambda: ops.gemm(a, b, b_ks=bks, out=out, variant=variant)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f()
            e1.record()
            torch.cuda.synchronize()
            res.append(f"{variant} {2.0 * M * N * K * 20 / e0.elapsed_time(e1) / 1e9:7.1f}")
        print(f"{name:24s} {fname:10s} " + " | ".join(res) + "  TFLOP/s", flush=True)
