"""A/B of the ViT attention forward forms (UNIMP_ATTN_VIT = 0 | 3 | 5 | 9, read once per process): time at the step's shape
(512 images x 16 heads x 257 x 64) and error against an fp32 softmax on a small batch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
torch.manual_seed(0)
bf = torch.bfloat16
B, H, S, D = 512, 16, 257, 64
qkv = torch.randn(B, S, H, 3 * D, device="cuda").to(bf)
q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
o, lse = ops.attn_fwd(q, k, v, D ** -0.5, 0, None, None, 0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    e0.record()
    for _ in range(10):
        ops.attn_fwd(q, k, v, D ** -0.5, 0, None, None, 0)
    e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
n = 8
qf, kf, vf = (t[:n].float() for t in (q, k, v))
s = torch.einsum("bqhd,bkhd->bhqk", qf, kf) * D ** -0.5
want = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), vf)
lse_want = torch.logsumexp(s, -1)
err = float((o[:n].float() - want).abs().max())
lerr = float((lse[:n, :, :S] - lse_want).abs().max())
path = "/tmp/vit_attn_ref.pt"
form = os.environ.get("UNIMP_ATTN_VIT", "default")
if form == "0":
    torch.save(o.cpu(), path); dref = 0.0
else:
    dref = float((o.cpu().float() - torch.load(path).float()).abs().max()) if os.path.exists(path) else float("nan")
print(f"UNIMP_ATTN_VIT={form}: {min(ts) * 1e3:.1f} us (median {sorted(ts)[2] * 1e3:.1f})  max|o - fp32| {err:.2e}  max|lse - fp32| {lerr:.2e}  max|o - form 0| {dref:.2e}", flush=True)
