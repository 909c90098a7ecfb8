"""ad-hoc: forward GEMM shapes of the step with the weight operand k-contiguous (W [N,K], the reference layout) vs k-strided
(a transposed copy W^T [K,N])."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
torch.manual_seed(0)
def run(name, m, n, k):
    x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    w = torch.randn(n, k, device="cuda").to(torch.bfloat16)
    wt = w.t().contiguous()
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    res = []
    for bks, b in ((False, w), (True, wt)):
        best = 0
        for variant in ("pp256", "w8", "pp128"):
            for _ in range(2): ops.gemm(x, b, b_ks=bks, out=out, variant=variant)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): ops.gemm(x, b, b_ks=bks, out=out, variant=variant)
            e1.record(); torch.cuda.synchronize()
            best = max(best, 2 * m * n * k * 8 / e0.elapsed_time(e1) / 1e9)
        res.append(best)
    print(f"{name:28s} M={m:6d} N={n:6d} K={k:6d}   W[N,K] {res[0]:7.1f}   W^T[K,N] {res[1]:7.1f} TFLOP/s  ({res[1] / res[0] - 1:+.1%})", flush=True)
run("LM up", 24576, 10240, 2560)
run("LM down", 24576, 2560, 10240)
run("LM qkv", 24576, 7680, 2560)
run("LM attn out", 24576, 2560, 2560)
run("ViT up", 98688, 4096, 1024)
run("ViT down", 98688, 1024, 4096)
run("ViT qkv", 98688, 3072, 1024)
run("ViT out", 98688, 1024, 1024)
run("head", 24576, 74053 // 8 * 8, 2560)
