"""Localises differences between the third-generation dK/dV kernel (attention3.hip; generation 2) and the first-generation one
(generation 4) on small cases: max |difference| per (32-key block, 16-column block) of dk and dv, relative to the tensor's scale."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops, _lib

bf = torch.bfloat16
CASES = [(1, 1, 32, 32, 0), (1, 1, 64, 64, 0), (1, 1, 64, 64, 1), (1, 1, 256, 256, 0), (1, 2, 512, 512, 1), (2, 2, 512, 512, 1)]


def run(B, H, Sq, Sk, mode, special=None):
    D = 80
    g = torch.Generator().manual_seed(Sq + mode)
    qkv = torch.randn(B, Sq, H, 3 * D, generator=g).to(bf).cuda()
    if special == "q1":          # Q = const: S identical for all rows
        qkv[..., :D] = 0.25
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    do = torch.randn(B, Sq, H, D, generator=g).to(bf).cuda()
    kv_len = None
    if B > 1:
        kv_len = torch.tensor([Sk, Sk - 37][:B], dtype=torch.int32).cuda()
    out = {}
    for gen in (4, 2):
        _lib.lib().unimp_attn_set_generation(gen)
        o, lse = ops.attn_fwd(q, k, v, D ** -0.5, mode, kv_len)
        dqkv = torch.zeros_like(qkv)
        dq, dk, dv = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
        ops.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, D ** -0.5, mode, kv_len)
        torch.cuda.synchronize()
        out[gen] = (dk.float().cpu().clone(), dv.float().cpu().clone())
    _lib.lib().unimp_attn_set_generation(2)
    print(f"== B{B} H{H} {Sq}x{Sk} mode{mode} {special or ''}")
    for name, i in (("dk", 0), ("dv", 1)):
        a, b = out[4][i], out[2][i]
        scale = a.abs().max().item() + 1e-9
        err = (a - b).abs() / scale                      # [B, S, H, D]
        print(f"  {name}: scale {scale:.3g}  max rel err {err.max().item():.3g}  nan {int(torch.isnan(b).sum())}")
        if err.max().item() > 2e-2 or torch.isnan(b).any():
            e = torch.nan_to_num(err, nan=99.0)
            for bb in range(B):
                for hh in range(H):
                    m = e[bb, :, hh, :].reshape(Sk // 32, 32, D // 16, 16).amax((1, 3))
                    print(f"   b{bb} h{hh} rows = 32-key blocks, cols = 16-d blocks:")
                    for r in range(m.shape[0]):
                        print("     " + " ".join(f"{x:7.2g}" for x in m[r].tolist()))
            # inside the worst key block: per key
            bb, ss, hh, dd = [int(x) for x in torch.unravel_index(e.argmax(), e.shape)]
            kb = ss // 32
            per_key = e[bb, kb * 32:(kb + 1) * 32, hh, :].amax(1)
            print(f"   worst block b{bb} h{hh} keys {kb * 32}..: per key " + " ".join(f"{x:.1g}" for x in per_key.tolist()))
            per_d = e[bb, kb * 32:(kb + 1) * 32, hh, :].amax(0)
            print("   per d " + " ".join(f"{x:.1g}" for x in per_d.tolist()))


for c in CASES:
    run(*c)
run(1, 1, 64, 64, 0, "q1")
