"""LM-shape attention (b = 64, 32 heads of 80, L = 512, causal, fills 75-100 %): the padded call with kv_len against the same sequences as
row ranges of a packed buffer (q_row_off / k_row_off); forward and backward, ms per call.  usage: [min_fill]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops     # noqa: E402

bf = torch.bfloat16
B, H, D, L = 64, 32, 80, 512
min_fill = float(sys.argv[1]) if len(sys.argv) > 1 else 0.75
g = torch.Generator().manual_seed(0)
lens = torch.randint(int(min_fill * L), L + 1, (B,), generator=g).to(torch.int32)
off = (torch.cumsum(lens, 0) - lens).to(torch.int32)
n = int(lens.sum())
M = (n + 2047) // 2048 * 2048
scale = D ** -0.5
qkv = torch.randn(B, L, H, 3 * D, generator=g).to(bf).cuda()
do = torch.randn(B, L, H, D, generator=g).to(bf).cuda()
q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
kvl = lens.cuda()
rows_of = torch.cat([b * L + torch.arange(int(lens[b])) for b in range(B)]).cuda()
qkv_p = torch.zeros(1, M, H, 3 * D, dtype=bf, device="cuda"); qkv_p[0, :n] = qkv.view(B * L, H, 3 * D)[rows_of]
do_p = torch.zeros(1, M, H, D, dtype=bf, device="cuda"); do_p[0, :n] = do.view(B * L, H, D)[rows_of]
qp, kp, vp = qkv_p[..., :D], qkv_p[..., D:2 * D], qkv_p[..., 2 * D:]
pr = ops.PackedRows(B, L, off.cuda(), kvl, n)
dqkv, dqkv_p = torch.empty_like(qkv), torch.zeros_like(qkv_p)


def timeit(f, reps=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


o, lse = ops.attn_fwd(q, k, v, scale, 1, kvl)
o_p, lse_p = ops.attn_fwd(qp, kp, vp, scale, 1, q_rows=pr, k_rows=pr)
t = {}
t["fwd padded"] = timeit(lambda: ops.attn_fwd(q, k, v, scale, 1, kvl))
t["fwd packed"] = timeit(lambda: ops.attn_fwd(qp, kp, vp, scale, 1, q_rows=pr, k_rows=pr))
t["bwd padded"] = timeit(lambda: ops.attn_bwd(q, k, v, o, lse, do, dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:], scale, 1, kvl))
t["bwd packed"] = timeit(lambda: ops.attn_bwd(qp, kp, vp, o_p, lse_p, do_p, dqkv_p[..., :D], dqkv_p[..., D:2 * D], dqkv_p[..., 2 * D:], scale, 1, q_rows=pr, k_rows=pr))
t["fwd padded, no kv_len (all 512 keys)"] = timeit(lambda: ops.attn_fwd(q, k, v, scale, 1))
print(f"fill {min_fill:.2f}-1.00: valid rows {n} of {B * L} ({n / (B * L):.3f}); sum len^2 / (B L^2) = {float((lens.float() ** 2).sum()) / (B * L * L):.3f}")
for kname, ms in t.items():
    print(f"  {kname:40s} {ms * 1e3:8.1f} us")
