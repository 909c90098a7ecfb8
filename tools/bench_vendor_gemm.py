"""ad-hoc, MEASUREMENT ONLY (not on the product path): this library's GEMM next to the vendor library behind torch.matmul
(hipBLASLt / rocBLAS) on the step's shapes, to place the 1.1-1.3 PF plateau.  y[M,N] = x[M,K] . w[N,K]^T, bf16."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
torch.manual_seed(0)


def timed(fn, flops, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return flops * reps / e0.elapsed_time(e1) / 1e9


def run(name, m, n, k):
    x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    w = torch.randn(n, k, device="cuda").to(torch.bfloat16)
    out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    fl = 2.0 * m * n * k
    ours = max(timed(lambda: ops.gemm(x, w, out=out, variant=v), fl) for v in ("pp256", "w8"))
    vend = timed(lambda: torch.matmul(x, w.t(), out=out), fl)
    wt = w.t().contiguous()
    vend_nn = timed(lambda: torch.matmul(x, wt, out=out), fl)
    print(f"{name:30s} M={m:6d} N={n:6d} K={k:6d}  ours {ours:7.1f}  vendor(NT) {vend:7.1f}  vendor(NN) {vend_nn:7.1f} TFLOP/s", flush=True)


run("square", 8192, 8192, 8192)
run("LM up-proj (b=48)", 22512, 10240, 2560)
run("LM down-proj", 22512, 2560, 10240)
run("LM qkv", 22512, 7680, 2560)
run("LM attn out", 22512, 2560, 2560)
run("ViT mlp up (384 images)", 98688, 4096, 1024)
run("ViT mlp down", 98688, 1024, 4096)
