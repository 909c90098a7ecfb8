"""ad-hoc: operand-form sweep at one shape (TFLOP/s per variant)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
torch.manual_seed(0)
def run(name, m, n, k, aks, bks, lda=None, ldb=None):
    sa = (k, lda or m) if aks else (m, lda or k)
    sb = (k, ldb or n) if bks else (n, ldb or k)
    a = torch.randn(sa, device="cuda").to(torch.bfloat16)[:, :(m if aks else k)]
    b = torch.randn(sb, device="cuda").to(torch.bfloat16)[:, :(n if bks else k)]
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    res = []
    for variant in ("pp256", "w8", "w4"):
        for _ in range(2): ops.gemm(a, b, a_ks=bool(aks), b_ks=bool(bks), out=out, variant=variant)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8): ops.gemm(a, b, a_ks=bool(aks), b_ks=bool(bks), out=out, variant=variant)
        e1.record(); torch.cuda.synchronize()
        res.append(f"{variant} {2*m*n*k*8/e0.elapsed_time(e1)/1e9:7.1f}")
    print(f"{name:34s} " + " | ".join(res), flush=True)
M = 8192
for aks, bks in ((0, 0), (0, 1), (1, 0), (1, 1)):
    run(f"sq 8192^3 aks={aks} bks={bks}", M, M, M, aks, bks)
run("fwd up 24576x10240x2560", 24576, 10240, 2560, 0, 0)
run("fwd down 24576x2560x10240", 24576, 2560, 10240, 0, 0)
run("vit fc1 98688x4096x1024", 98688, 4096, 1024, 0, 0)
run("dW-like 10240x2560 K=12288 (1,1)", 10240, 2560, 12288, 1, 1)
run("same, lda padded +64", 10240, 2560, 12288, 1, 1, lda=10240 + 64, ldb=2560 + 64)
run("dW-like 10240x2560 K=12288 (1,0)", 10240, 2560, 12288, 1, 0)
run("dW-like 10240x2560 K=12288 (0,0)", 10240, 2560, 12288, 0, 0)
