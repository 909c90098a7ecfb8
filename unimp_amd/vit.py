"""CLIP ViT vision tower on the HIP kernels (forward only: the tower is frozen and runs under no_grad in
Flamingo._encode_vision_x).  Parameter names follow open_clip's VisionTransformer
so OpenFlamingo / open_clip checkpoints load; math follows UniMP/xformers_model/clip.py:50-206 with the fused
in_proj of nn.MultiheadAttention and the open_clip output selection (tokens = last block, CLS dropped, no ln_post).
"""
import torch
import torch.nn as nn

from . import ops

bf16 = torch.bfloat16

# open_clip model configs used by UniMP (mmrec.py:477 uses ViT-L-14 / openai)
VISION_CONFIGS = {
    "ViT-L-14": dict(image_size=224, patch_size=14, width=1024, layers=24, heads=16, mlp_dim=4096, output_dim=768),
    "ViT-B-32": dict(image_size=224, patch_size=32, width=768, layers=12, heads=12, mlp_dim=3072, output_dim=512),
    "ViT-B-16": dict(image_size=224, patch_size=16, width=768, layers=12, heads=12, mlp_dim=3072, output_dim=512),
}


class _MHAParams(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.randn(3 * d, d) * d ** -0.5)
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)


class _MLPParams(nn.Module):
    def __init__(self, d, m):
        super().__init__()
        self.c_fc = nn.Linear(d, m)
        self.c_proj = nn.Linear(m, d)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d, heads, mlp, act):
        super().__init__()
        self.ln_1 = nn.LayerNorm(d)
        self.attn = _MHAParams(d)
        self.ln_2 = nn.LayerNorm(d)
        self.mlp = _MLPParams(d, mlp)
        self.heads, self.act = heads, act

    def forward(self, h, N, S):
        """h: [N*S, D] bf16 -> [N*S, D]"""
        from . import functional as F_
        D = h.shape[1]
        hd = D // self.heads
        ws = (self.attn.in_proj_weight, self.attn.out_proj.weight, self.mlp.c_fc.weight, self.mlp.c_proj.weight)
        mx = all(F_._mx_ok(w, h.shape[0]) for w in ws)          # opt-in MX-fp8 path for the frozen tower (functional.FP8_FROZEN)

        def lin(x, w, attn=False, **kw):
            if mx:
                return ops.gemm_mx(ops.mx_quantize(x), F_._frozen_mx(w), **kw)
            if (F_.FROZEN_WT_ATTN if attn else F_.FROZEN_WT) and not w.requires_grad and x.shape[0] > 64:
                return ops.gemm(x, F_._frozen_t(w), b_ks=True, **kw)        # W^T copy: whole rows per LDS-DMA instruction, same bits
            return ops.gemm(x, w, b_pk=F_._frozen_pk(w), **kw)
        a, _, _ = ops.layernorm_fwd(h, self.ln_1.weight, self.ln_1.bias, self.ln_1.eps)
        qkv = lin(a, self.attn.in_proj_weight, attn=True, bias=self.attn.in_proj_bias).view(N, S, 3, self.heads, hd)
        o, _ = ops.attn_fwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], hd ** -0.5, ops.MASK_NONE)
        h = lin(o.view(N * S, D), self.attn.out_proj.weight, attn=True, bias=self.attn.out_proj.bias, res=h)
        a, _, _ = ops.layernorm_fwd(h, self.ln_2.weight, self.ln_2.bias, self.ln_2.eps)
        f = lin(a, self.mlp.c_fc.weight, bias=self.mlp.c_fc.bias, act=self.act)
        return lin(f, self.mlp.c_proj.weight, bias=self.mlp.c_proj.bias, res=h)


class _Transformer(nn.Module):
    def __init__(self, d, layers, heads, mlp, act):
        super().__init__()
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(d, heads, mlp, act) for _ in range(layers)])


class VisionTransformer(nn.Module):
    def __init__(self, image_size=224, patch_size=14, width=1024, layers=24, heads=16, mlp_dim=4096, output_dim=768,
                 quick_gelu=True):
        super().__init__()
        self.image_size, self.patch_size, self.width = image_size, patch_size, width
        self.output_tokens = False
        n = (image_size // patch_size) ** 2
        self.conv1 = nn.Conv2d(3, width, patch_size, patch_size, bias=False)
        self.class_embedding = nn.Parameter(torch.randn(width) * width ** -0.5)
        self.positional_embedding = nn.Parameter(torch.randn(n + 1, width) * width ** -0.5)
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = _Transformer(width, layers, heads, mlp_dim, "quick_gelu" if quick_gelu else "gelu")
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(torch.randn(width, output_dim) * width ** -0.5)
        self._wcache = None

    def _conv_weight_matrix(self):
        """conv1.weight [D,3,P,P] as a zero-padded GEMM operand [D, roundup8(3*P*P)]; rebuilt when the weight changes."""
        w = self.conv1.weight
        key = (w.data_ptr(), w._version)
        if self._wcache is None or self._wcache[0] != key:
            K = w[0].numel()
            ld = (K + 7) // 8 * 8
            m = torch.zeros((w.shape[0], ld), dtype=bf16, device=w.device)
            m[:, :K].copy_(w.detach().reshape(w.shape[0], K))
            self._wcache = (key, m)
        return self._wcache[1]

    def forward(self, x):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("unimp_amd ViT is forward-only (frozen tower under no_grad, as in open_flamingo)")
        N = x.shape[0]
        P, D = self.patch_size, self.width
        wm = self._conv_weight_matrix()
        cols = ops.vit_patchify(x.contiguous(), P, wm.shape[1])
        patch = ops.gemm(cols, wm)                                                   # conv1 as GEMM (clip.py:77-80)
        g2 = patch.shape[0] // N
        S = g2 + 1
        xt = ops.vit_assemble(patch, self.class_embedding, self.positional_embedding, N, g2)
        h, _, _ = ops.layernorm_fwd(xt.view(N * S, D), self.ln_pre.weight, self.ln_pre.bias, self.ln_pre.eps)
        for blk in self.transformer.resblocks:
            h = blk(h, N, S)
        h3 = h.view(N, S, D)
        pooled, _, _ = ops.layernorm_fwd(h3[:, 0], self.ln_post.weight, self.ln_post.bias, self.ln_post.eps)
        pooled = ops.gemm(pooled, self.proj, b_ks=True)
        tokens = h3[:, 1:]
        return (pooled, tokens) if self.output_tokens else pooled


class CLIPStub(nn.Module):
    """Holder with a ``.visual`` attribute like an open_clip CLIP model (Flamingo takes ``vision_encoder.visual``)."""

    def __init__(self, visual):
        super().__init__()
        self.visual = visual
