"""Do all GEMM kernel variants give the same BITS on a forward (k-contiguous) problem?  The per-sample-independence tests rely on it:
the variant is chosen per (M, N, K), so a sample decoded alone (M = L) and inside a batch (M = B L) may run different variants."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops       # noqa: E402

bf = torch.bfloat16
torch.manual_seed(0)
shapes = [(512, 4096, 4096), (1024, 4096, 4096), (512, 12288, 4096), (1024, 16384, 4096), (1024, 4096, 16384), (512, 74053, 4096), (1024, 2560, 2560),
          (1024, 1024, 1024), (2056, 1024, 1024),
          # the vision path at 8 / 16 images: ViT-L (257 tokens per image) and the Perceiver (320 kv rows, 64 latents per image)
          (2056, 3072, 1024), (4112, 3072, 1024), (2056, 4096, 1024), (4112, 4096, 1024), (2056, 1024, 4096), (4112, 1024, 4096), (4112, 1024, 1024),
          (2560, 1024, 1024), (5120, 1024, 1024), (512, 512, 1024), (1024, 512, 1024), (512, 1024, 512), (512, 4096, 1024), (1024, 4096, 1024),
          (512, 1024, 4096), (1024, 1024, 4096), (2048, 602, 1024),
          (512, 2560, 512), (1024, 2560, 512), (512, 2560, 10240), (1024, 2560, 10240)]      # the gated cross-attention
for M, N, K in shapes:
    a = torch.randn(M, K, device="cuda").to(bf)
    b = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
    res = torch.randn(M, N, device="cuda").to(bf)
    bias = torch.randn(N, device="cuda").to(bf)
    gate = torch.tensor([0.5], dtype=bf, device="cuda")
    for epi in ("plain", "res", "gelu", "bias", "bias+quick_gelu", "bias+res", "gate+res", "bias+gate+res"):
        kw = {"plain": {}, "res": dict(res=res), "gelu": dict(act="gelu"), "bias": dict(bias=bias), "bias+quick_gelu": dict(bias=bias, act="quick_gelu"),
              "bias+res": dict(bias=bias, res=res), "gate+res": dict(res=res, gate=gate), "bias+gate+res": dict(bias=bias, res=res, gate=gate)}[epi]
        outs = {}
        for v in ("v1", "dma256", "dma128", "pp256", "pp128", "w8", "pp256p", "pp256x", "pp256a", "pp128a", "pp256b", "w4x", "w4x_pf"):
            try:
                ldc = (N + 7) // 8 * 8
                o = ops.gemm(a, b, variant=v, ldc=ldc, **kw)
                outs[v] = o.clone()
            except Exception as e:       # noqa: BLE001
                outs[v] = None
        ref = outs["v1"]
        diff = {v: (None if o is None else int((o.view(torch.int16) != ref.view(torch.int16)).sum())) for v, o in outs.items()}
        bad = {v: n for v, n in diff.items() if n}
        print(f"M={M:5d} N={N:6d} K={K:6d} {epi:16s} " + ("all variants bit-identical" if not bad else f"DIFFER from v1: {bad}"), flush=True)
