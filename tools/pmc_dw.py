"""PMC workload for the two-workgroups-per-CU GEMM (variant dw / dwpk) next to pp256a on one ViT and one LM shape (separate rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
bf = torch.bfloat16
torch.manual_seed(0)
for (M, N, K) in ((131584, 4096, 1024), (32768, 2560, 2560)):
    a = torch.randn(M, K, device="cuda").to(bf)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
    wt = w.t().contiguous()
    pk = ops.pack_b(w, False)
    bias = torch.randn(N, device="cuda").to(bf)
    out = torch.empty(M, N, dtype=bf, device="cuda")
    for _ in range(4):
        ops.gemm(a, wt, b_ks=True, bias=bias, out=out, variant="pp256a")
        ops.gemm(a, w, bias=bias, out=out, variant="dwpk", b_pk=pk)
        ops.gemm(a, wt, b_ks=True, bias=bias, out=out, variant="dw")
    torch.cuda.synchronize()
