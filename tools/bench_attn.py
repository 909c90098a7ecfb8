"""ad-hoc: attention forward / backward time of the causal LM shape (hd 80, fused-QKV strides) vs sequence length at a fixed
token count, to separate per-block fixed cost from per-tile cost."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
torch.manual_seed(0)
bf16 = torch.bfloat16


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def run(B, L, H=32, D=80, mask=ops.MASK_CAUSAL):
    qkv = torch.randn(B, L, H, 3 * D, device="cuda").to(bf16)
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    scale = D ** -0.5
    o, lse = ops.attn_fwd(q, k, v, scale, mask)
    do = torch.randn_like(o)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
    tf = timed(lambda: ops.attn_fwd(q, k, v, scale, mask, out=o))
    tb = timed(lambda: ops.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, scale, mask))
    fl = 4.0 * B * H * L * L * D * (0.5 if mask == ops.MASK_CAUSAL else 1.0)
    print(f"B={B:4d} L={L:5d} hd={D}: fwd {tf:8.1f} us ({fl / tf / 1e6:6.1f} TF/s)   bwd {tb:8.1f} us ({2.5 * fl / tb / 1e6:6.1f} TF/s)", flush=True)


run(48, 469)
run(48, 512)
run(24, 1024)
run(12, 2048)
run(6, 4096)
run(48, 469, D=64)
run(48, 469, D=128)
run(48, 512, mask=ops.MASK_NONE)
