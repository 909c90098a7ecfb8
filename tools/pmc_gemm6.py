"""PMC workload: three launches of the persistent ping-pong kernel on the step's dominant forward shape, for rocprofv3 --pmc passes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
torch.manual_seed(0)
m, n, k = 24576, 10240, 2560
a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
b = torch.randn(n, k, device="cuda").to(torch.bfloat16)
out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    ops.gemm(a, b, out=out, variant="pp256p")
torch.cuda.synchronize()
