"""LayerNorm fwd / bwd and a plain device copy at the step's shapes: achieved TB/s (algorithmic bytes).
--rotate N: cycle through N distinct operand sets (N x 4 buffers > the 256 MB Infinity Cache) so every launch streams from HBM,
as the kernels do inside the training step; without it the same buffers are re-read and part of them is served by the cache."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
bf = torch.bfloat16
ROT = int(sys.argv[sys.argv.index("--rotate") + 1]) if "--rotate" in sys.argv else 1
def timeit(fn, n=24):
    for i in range(ROT): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i % ROT)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
for R, D in ((32768, 2560), (131584, 1024)):
    S = []
    for i in range(ROT):
        x = torch.randn(R, D, device="cuda").to(bf); dy = torch.randn_like(x); y = torch.empty_like(x)
        S.append((x, dy, y))
    g = torch.randn(D, device="cuda").to(bf); b = torch.zeros_like(g)
    nb = R * D * 2
    t = timeit(lambda i: S[i][2].copy_(S[i][0])); print(f"[{R}x{D}] torch copy        {t*1e3:7.1f} us  {2*nb/t/1e9:.2f} TB/s")
    t = timeit(lambda i: ops.add(S[i][0], S[i][1], out=S[i][2])); print(f"[{R}x{D}] add (2r+1w)       {t*1e3:7.1f} us  {3*nb/t/1e9:.2f} TB/s")
    yy, mean, rstd = ops.layernorm_fwd(S[0][0], g, b, 1e-5)
    t = timeit(lambda i: ops.layernorm_fwd(S[i][0], g, b, 1e-5, out=S[i][2])); print(f"[{R}x{D}] ln_fwd            {t*1e3:7.1f} us  {2*nb/t/1e9:.2f} TB/s")
    t = timeit(lambda i: ops.layernorm_bwd(S[i][1], S[i][0], g, mean, rstd, want_wgrad=False)); print(f"[{R}x{D}] ln_bwd            {t*1e3:7.1f} us  {3*nb/t/1e9:.2f} TB/s")
    t = timeit(lambda i: ops.layernorm_bwd(S[i][1], S[i][0], g, mean, rstd, dres=S[i][2], want_wgrad=False)); print(f"[{R}x{D}] ln_bwd (+dres)    {t*1e3:7.1f} us  {4*nb/t/1e9:.2f} TB/s")
    t = timeit(lambda i: ops.layernorm_bwd(S[i][1], S[i][0], g, mean, rstd, dres=S[i][2], want_wgrad=True)); print(f"[{R}x{D}] ln_bwd wgrad      {t*1e3:7.1f} us  {4*nb/t/1e9:.2f} TB/s")
    del S
