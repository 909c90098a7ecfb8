import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unimp_amd import ops
torch.manual_seed(0)
for rows, D in [(300, 4096), (129, 2560), (70, 1024), (33, 96)]:
    x = (torch.randn(rows, D) * 2).to(torch.bfloat16).cuda()
    g = (1 + 0.1 * torch.randn(D)).to(torch.bfloat16).cuda()
    y, mean, rstd = ops.layernorm_fwd(x, g, None, 1e-5)
    want = ops.mx_quantize(y)
    got, m2, r2 = ops.layernorm_fwd_mx(x, g, None, 1e-5)
    bad = (got.q != want.q).nonzero()
    print(rows, D, "scales equal", torch.equal(got.scales, want.scales), "stats equal", torch.equal(mean, m2), torch.equal(rstd, r2), "mismatching elements", bad.shape[0], "of", rows * D)
    if bad.shape[0]:
        print(" first", bad[:8].tolist(), "cols mod 8:", sorted(set((bad[:, 1] % 8).tolist())), "rows:", sorted(set(bad[:, 0].tolist()))[:10])
        i, j = bad[0].tolist()
        print(" got", got.q[i, j - j % 8:j - j % 8 + 8].tolist(), "want", want.q[i, j - j % 8:j - j % 8 + 8].tolist(), "y", y[i, j - j % 8:j - j % 8 + 8].tolist(), "scale", int(want.scales[i, j // 32]))
