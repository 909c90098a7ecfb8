"""ad-hoc: per-tile fixed cost of the large-tile kernels on the ViT shapes: time vs K at fixed M, N, per variant and epilogue."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
torch.manual_seed(0)
M = 98688
def t(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 6 * 1e3
for N in (int(a) for a in (sys.argv[1:] or ("1024", "4096"))):
    res = torch.randn(M, N, device="cuda").to(torch.bfloat16); bias = torch.randn(N, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for K in (512, 1024, 2048, 4096):
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
        line = []
        for variant in ("pp256", "pp128", "w8"):
            a = t(lambda: ops.gemm(x, w, out=out, variant=variant))
            b = t(lambda: ops.gemm(x, w, out=out, variant=variant, bias=bias, res=res))
            c = t(lambda: ops.gemm(x, w, out=out, variant=variant, bias=bias, act="quick_gelu"))
            line.append(f"{variant}: plain {a:7.1f} us ({2*M*N*K/a/1e6:6.0f} TF)  +bias+res {b:7.1f}  +bias+qgelu {c:7.1f}")
        print(f"N={N} K={K}: " + " | ".join(line), flush=True)
