"""do all kernel variants sum a k-strided B operand in the same order?  (forward GEMMs reading a transposed weight copy must give the
same bits whichever variant a batch size tunes to)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
bf = torch.bfloat16
torch.manual_seed(0)
for M, N, K in ((1024, 512, 2560), (1536, 2560, 2560), (2048, 1024, 4096), (1024, 10240, 2560), (1280, 2560, 10240)):
    a = torch.randn(M, K, device="cuda").to(bf)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
    wt = w.t().contiguous()
    bias = torch.randn(N, device="cuda").to(bf)
    res = torch.randn(M, N, device="cuda").to(bf)
    for name, kw in (("plain", {}), ("bias+gelu", dict(bias=bias, act="gelu")), ("bias+res", dict(bias=bias, res=res))):
        ref_kc = ops.gemm(a, w, variant="pp256", **kw)
        outs = {v: ops.gemm(a, wt, b_ks=True, variant=v, **kw) for v in ("v1", "pp256", "pp128", "w8", "pp256p", "pp256x", "pp128x", "pp256a", "pp128a", "pp256b", "w4x", "w4x_pf", "dma256", "dma128")}
        base = outs["pp256"]
        diff = {v: int((o != base).sum()) for v, o in outs.items()}
        print(f"[{M},{N},{K}] {name:10s} elements differing from pp256(W^T): {diff};  W^T vs W (pp256): {int((base != ref_kc).sum())} of {base.numel()}", flush=True)
