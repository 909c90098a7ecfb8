"""Oracle: CLIP ViT vision tower (fp32, CPU).  TEST INFRASTRUCTURE ONLY.

Math follows the in-tree UniMP/xformers_model/clip.py:
  * patch embedding + class token + position embedding   clip.py:50-85
  * pre-LN encoder block, MHA scale hd^-0.5               clip.py:88-141,159-206
  * MLP fc1 -> QuickGELU -> fc2                            clip.py:144-156
  * pre_layrnorm before the encoder                        clip.py:423,460
Parameter naming / fused in_proj / output selection follow open_clip's
``VisionTransformer`` (the production tower, UniMP/mmrec.py:477-478):
returns ``(pooled, tokens)``; tokens = last block output without CLS and WITHOUT ln_post.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


from .numerics import st


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


class _MHA(nn.Module):
    """nn.MultiheadAttention-compatible parameter layout (in_proj_weight ordered q,k,v)."""

    def __init__(self, d, heads):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.randn(3 * d, d) * d ** -0.5)
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)
        self.heads = heads

    def forward(self, x):
        B, S, D = x.shape
        H = self.heads
        qkv = st("gemm", F.linear(x, self.in_proj_weight, self.in_proj_bias))
        q, k, v = qkv.view(B, S, 3, H, D // H).permute(2, 0, 3, 1, 4)
        att = (q @ k.transpose(-1, -2)) * (D // H) ** -0.5
        att = st("attn_p", att.softmax(-1))
        o = st("attn_o", (att @ v).transpose(1, 2).reshape(B, S, D))
        return self.out_proj(o)


class _MLP(nn.Module):
    def __init__(self, d, m, act):
        super().__init__()
        self.c_fc = nn.Linear(d, m)
        self.c_proj = nn.Linear(m, d)
        self.act = act

    def forward(self, x):
        return self.c_proj(st("act", self.act(self.c_fc(x))))


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d, heads, mlp, act):
        super().__init__()
        self.ln_1 = nn.LayerNorm(d)
        self.attn = _MHA(d, heads)
        self.ln_2 = nn.LayerNorm(d)
        self.mlp = _MLP(d, mlp, act)

    def forward(self, x):
        x = st("res", x + self.attn(st("ln", self.ln_1(x))))
        return st("res", x + self.mlp(st("ln", self.ln_2(x))))


class _Transformer(nn.Module):
    def __init__(self, d, layers, heads, mlp, act):
        super().__init__()
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(d, heads, mlp, act) for _ in range(layers)])

    def forward(self, x):
        for b in self.resblocks:
            x = b(x)
        return x


class VisionTransformer(nn.Module):
    def __init__(self, image_size=224, patch_size=14, width=1024, layers=24, heads=16,
                 mlp_dim=4096, output_dim=768, quick_gelu_act=True):
        super().__init__()
        self.image_size, self.patch_size, self.width = image_size, patch_size, width
        self.output_tokens = True
        n = (image_size // patch_size) ** 2
        self.conv1 = nn.Conv2d(3, width, patch_size, patch_size, bias=False)
        self.class_embedding = nn.Parameter(torch.randn(width) * width ** -0.5)
        self.positional_embedding = nn.Parameter(torch.randn(n + 1, width) * width ** -0.5)
        self.ln_pre = nn.LayerNorm(width)
        act = quick_gelu if quick_gelu_act else F.gelu
        self.transformer = _Transformer(width, layers, heads, mlp_dim, act)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(torch.randn(width, output_dim) * width ** -0.5)

    def forward(self, x):
        x = st("gemm", self.conv1(x))                       # (B, D, g, g)          clip.py:77-80
        x = x.flatten(2).transpose(1, 2)                    # (B, n, D)
        cls = self.class_embedding.expand(x.shape[0], 1, -1)
        x = st("res", torch.cat([cls, x], 1) + self.positional_embedding)   # clip.py:82-84
        x = st("res", self.ln_pre(x))                       # clip.py:460
        x = self.transformer(x)
        pooled, tokens = x[:, 0], x[:, 1:]
        pooled = self.ln_post(pooled) @ self.proj           # ln_post on pooled ONLY (SURVEY a-5)
        return pooled, tokens
