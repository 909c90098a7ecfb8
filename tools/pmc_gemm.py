"""PMC workload: a few launches of the dominant GEMM shapes (ping-pong kernel) for rocprofv3 --pmc passes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
torch.manual_seed(0)
M = 24576
for (m, n, k, aks, bks) in [(M, 10240, 2560, 0, 0), (M, 2560, 10240, 0, 0), (M, 2560, 10240, 0, 1), (10240, 2560, M, 1, 1)]:
    a = torch.randn((k, m) if aks else (m, k), device="cuda").to(torch.bfloat16)
    b = torch.randn((k, n) if bks else (n, k), device="cuda").to(torch.bfloat16)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        ops.gemm(a, b, a_ks=bool(aks), b_ks=bool(bks), out=out, variant="pp256")
    torch.cuda.synchronize()
