"""LayerNorm fwd / bwd and a plain device copy at the step's shapes: achieved TB/s (algorithmic bytes)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
bf = torch.bfloat16
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
for R, D in ((24576, 2560), (98688, 1024)):
    x = torch.randn(R, D, device="cuda").to(bf); dy = torch.randn_like(x); g = torch.randn(D, device="cuda").to(bf); b = torch.zeros_like(g)
    y = torch.empty_like(x)
    nb = R * D * 2
    t = timeit(lambda: y.copy_(x)); print(f"[{R}x{D}] torch copy        {t*1e3:7.1f} us  {2*nb/t/1e9:.2f} TB/s")
    t = timeit(lambda: ops.add(x, dy, out=y)); print(f"[{R}x{D}] add (2r+1w)       {t*1e3:7.1f} us  {3*nb/t/1e9:.2f} TB/s")
    yy, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5)
    t = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-5, out=y)); print(f"[{R}x{D}] ln_fwd            {t*1e3:7.1f} us  {2*nb/t/1e9:.2f} TB/s")
    t = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dres=dy, want_wgrad=False)); print(f"[{R}x{D}] ln_bwd (+dres)    {t*1e3:7.1f} us  {4*nb/t/1e9:.2f} TB/s")
    t = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dres=dy, want_wgrad=True)); print(f"[{R}x{D}] ln_bwd wgrad      {t*1e3:7.1f} us  {4*nb/t/1e9:.2f} TB/s")
