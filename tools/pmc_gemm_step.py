"""PMC workload: the step's dominant GEMM shapes at the bench batch (default 64 x 512 tokens), dispatched as the step dispatches them
(committed autotune table), three launches each, for separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | MFMA busy + GRBM)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
torch.manual_seed(0)
M = int(os.environ.get("PMC_M", 64 * 512))
for (n, k, bks) in [(10240, 2560, 0), (2560, 10240, 0), (10240, 2560, 1), (2560, 10240, 1)]:
    a = torch.randn(M, k, device="cuda").to(torch.bfloat16)
    b = torch.randn((k, n) if bks else (n, k), device="cuda").to(torch.bfloat16)
    out = torch.empty((M, n), dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        ops.gemm(a, b, b_ks=bool(bks), out=out)
    torch.cuda.synchronize()
