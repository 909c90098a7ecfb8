import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
torch.manual_seed(0)
for K in (32, 64, 96, 128, 160, 192, 256, 512):
    M = N = 256
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    want = a.float() @ b.float().t()
    got = ops.gemm(a, b, variant="w4").float()
    err = (got - want).abs()
    blk = err.view(16, 16, 16, 16).amax(dim=(1, 3))       # [m-tile, n-tile]
    bad = (blk > 0.05 * want.abs().max()).nonzero()
    print(f"K={K:4d} nh={K // 32}: max err {err.max().item():8.3f}  bad 16x16 tiles: {len(bad)}  first {bad[:6].tolist()}")
    if K == 192:
        # which k-slices are missing? compare against partial sums
        for h in range(K // 32):
            part = want - a[:, h * 32:(h + 1) * 32].float() @ b[:, h * 32:(h + 1) * 32].float().t()
            print("   minus slice", h, "err", (got - part).abs().max().item())
