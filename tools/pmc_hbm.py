"""PMC workload for the HBM-bound kernels: one launch group per kernel at step sizes, for separate rocprofv3 --pmc passes
(FETCH_SIZE | WRITE_SIZE) with --kernel-trace.  gfx950: FETCH_SIZE [KB] reports half of a wide coalesced stream (x2),
WRITE_SIZE [KB] is exact (MI355X_MICROARCH.md, HBM section)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
torch.manual_seed(0)
bf = torch.bfloat16
R, D = 24576, 2560
x = torch.randn(R, D, device="cuda").to(bf); dy = torch.randn_like(x); g = torch.randn(D, device="cuda").to(bf); b = torch.zeros_like(g)
for _ in range(2):
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5)
    ops.layernorm_bwd(dy, x, g, mean, rstd, dres=dy, want_wgrad=False)
    ops.layernorm_bwd(dy, x, g, mean, rstd, dres=dy, want_wgrad=True)
# decode-row GEMM (weights streamed once)
a = torch.randn(10, 2560, device="cuda").to(bf)
ws = [torch.randn(10240, 2560, device="cuda").to(bf) for _ in range(8)]
for w in ws:
    ops.gemm(a, w, variant="skinny")
# optimizer
n = 1_340_000_000 // 4
master = torch.randn(n, device="cuda"); m = torch.zeros_like(master); v = torch.zeros_like(master)
p = master.to(bf); gr = torch.randn(n, device="cuda").to(bf); ss = torch.zeros(1025, device="cuda")
for _ in range(2):
    ss.zero_(); ops.sumsq(gr, ss)
    ops.adamw_flat(master, m, v, p, gr, n // 2, 2e-4, 0.9, 0.999, 1e-8, 0.1, 1, ss, 1.0, 1.0, False)
torch.cuda.synchronize()
