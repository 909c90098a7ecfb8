"""PMC workload for the attention kernels at the step's shapes (b = 48 per GPU), for separate rocprofv3 --pmc passes with
--kernel-trace: LM causal (32 heads of 80, S = 512), ViT non-causal (16 heads of 64, S = 257, 384 images), gated cross
attention (8 heads of 64, 512 queries x 512 segment-masked keys), Perceiver (8 heads of 64, 64 x 320), MPT (hd 128, S = 1024)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
torch.manual_seed(0)
bf = torch.bfloat16
REP = int(os.environ.get("REP", 3))


def run(B, H, Sq, Sk, D, mode, bwd=True):
    if Sq == Sk:
        qkv = torch.randn(B, Sq, H, 3 * D, device="cuda").to(bf)
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
    else:
        q = torch.randn(B, Sq, H, D, device="cuda").to(bf)
        kv = torch.randn(B, Sk, 2, H, D, device="cuda").to(bf)
        k, v = kv[:, :, 0], kv[:, :, 1]
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        dk, dv = dkv[:, :, 0], dkv[:, :, 1]
    seg, seg_len, kv_len = None, 0, None
    if mode == 2:
        seg_len = 64
        seg = (torch.arange(Sq, device="cuda", dtype=torch.int32) * (Sk // 64) // Sq + 1).clamp(max=Sk // 64)[None].expand(B, -1).contiguous()
    if mode == 1:
        kv_len = torch.randint(int(0.75 * Sk), Sk + 1, (B,), device="cuda", dtype=torch.int32)
    do = torch.randn(B, Sq, H, D, device="cuda").to(bf)
    for _ in range(REP):
        o, lse = ops.attn_fwd(q, k, v, D ** -0.5, mode, kv_len, seg, seg_len)
        if bwd:
            ops.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, D ** -0.5, mode, kv_len, seg, seg_len)
    torch.cuda.synchronize()


which = os.environ.get("ATTN", "lm,vit,xattn,perc,mpt").split(",")
if "lm" in which:
    run(48, 32, 512, 512, 80, 1)
if "vit" in which:
    run(384, 16, 257, 257, 64, 0, bwd=False)
if "xattn" in which:
    run(48, 8, 512, 512, 64, 2)
if "perc" in which:
    run(384, 8, 64, 320, 64, 0)
if "mpt" in which:
    run(8, 32, 1024, 1024, 128, 1)
