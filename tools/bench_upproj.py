"""the two heavy-epilogue MLP GEMMs of the LM at the step's shape: up-projection + GELU + stored GELU' (two outputs) and the dX
through the activation (reads the stored derivative); TFLOP/s per kernel variant."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
bf = torch.bfloat16
M, N, K = 32768, 10240, 2560
x = torch.randn(M, K, device="cuda").to(bf); w = (torch.randn(N, K, device="cuda") * 0.02).to(bf); b = torch.zeros(N, device="cuda").to(bf)
y = torch.empty(M, N, dtype=bf, device="cuda"); pre = torch.empty(M, N, dtype=bf, device="cuda")
dy = torch.randn(M, K, device="cuda").to(bf)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
fl = 2.0 * M * N * K
for v in ("pp256", "w8", "pp256p"):
    t0 = timeit(lambda: ops.gemm(x, w, bias=b, out=y, variant=v))
    t1 = timeit(lambda: ops.gemm(x, w, bias=b, act="gelu", out=y, variant=v))
    t2 = timeit(lambda: ops.gemm(x, w, bias=b, act="gelu", pre=pre, pre_deriv=True, out=y, variant=v))
    print(f"{v:7s} bias {fl/t0/1e9:6.0f} TF | +gelu {fl/t1/1e9:6.0f} | +gelu +stored gelu' {fl/t2/1e9:6.0f} TF ({t2*1e3:.0f} us)", flush=True)
