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
    wt = w.t().contiguous()
    # ours: the forms the step uses -- the weight as stored ([out, in]: trainable layers) and read from its transposed copy (frozen layers)
    ours = {v: timed(lambda: ops.gemm(x, w, out=out, variant=v), fl) for v in ("pp256a", "w4x", "w4x_pf")}
    ours_t = {v: timed(lambda: ops.gemm(x, wt, b_ks=True, out=out, variant=v), fl) for v in ("pp256a", "pp256b", "w4x", "w4x_pf")}
    vend = timed(lambda: torch.matmul(x, w.t(), out=out), fl)
    vend_nn = timed(lambda: torch.matmul(x, wt, out=out), fl)
    bo, bt = max(ours, key=ours.get), max(ours_t, key=ours_t.get)
    best = max(ours[bo], ours_t[bt])
    print(f"{name:30s} M={m:6d} N={n:6d} K={k:6d}  ours [out,in] {ours[bo]:7.1f} ({bo})  ours W^T {ours_t[bt]:7.1f} ({bt})  vendor(NT) {vend:7.1f}  vendor(NN) {vend_nn:7.1f} TFLOP/s"
          f"   best ours / best vendor {best / max(vend, vend_nn):.3f}", flush=True)


print("# plain GEMMs (no epilogue), N(0,1) operands, TFLOP/s; the step's row counts (b = 64: 32 768 tokens / 131 584 ViT rows) and round 4's (b = 48: a ragged tile count)", flush=True)
run("square", 8192, 8192, 8192)
for tag, ml, mv in (("b=64", 32768, 131584), ("b=48", 22512, 98688)):
    run(f"LM up-proj ({tag})", ml, 10240, 2560)
    run(f"LM down-proj ({tag})", ml, 2560, 10240)
    run(f"LM qkv ({tag})", ml, 7680, 2560)
    run(f"LM attn out ({tag})", ml, 2560, 2560)
    run(f"ViT mlp up ({tag})", mv, 4096, 1024)
    run(f"ViT mlp down ({tag})", mv, 1024, 4096)
