"""Oracle: the in-tree Llama blocks (fp32, CPU).  TEST INFRASTRUCTURE ONLY.

Restates UniMP/xformers_model/llama.py:
  RMSNorm            llama.py:101-118  (variance in fp32, x*rsqrt(var+eps), * weight)
  rope tables/apply  llama.py:121-182  (inv_freq = base^(-2i/d), emb = cat(freqs,freqs), half-split)
  SwiGLU MLP         llama.py:185-199  (down(silu(gate(x)) * up(x)))
  causal attention   llama.py:202-308  (training path: LowerTriangularMask only -- padding masks are
                                        ignored)
  decoder layer/LM   llama.py:311-380, 709-880 (pre-norm residual, final norm, lm_head, shifted CE)
PINNED against the in-tree module (tests/golden/llama_tiny.npz, oracle/make_golden.py).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class RMSNorm(nn.Module):
    def __init__(self, d, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.eps = eps

    def forward(self, x):
        var = x.float().pow(2).mean(-1, keepdim=True)
        return self.weight * (x * torch.rsqrt(var + self.eps))


def rope_tables(L, d, base=10000.0):
    inv = 1.0 / (base ** (torch.arange(0, d, 2).float() / d))
    fr = torch.arange(L).float()[:, None] * inv[None]
    emb = torch.cat((fr, fr), -1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), -1)


class LlamaAttention(nn.Module):
    def __init__(self, d, heads):
        super().__init__()
        self.nh, self.hd = heads, d // heads
        self.q_proj = nn.Linear(d, d, bias=False)
        self.k_proj = nn.Linear(d, d, bias=False)
        self.v_proj = nn.Linear(d, d, bias=False)
        self.o_proj = nn.Linear(d, d, bias=False)

    def forward(self, x):
        B, L, D = x.shape
        sh = lambda t: t.view(B, L, self.nh, self.hd).transpose(1, 2)
        q, k, v = sh(self.q_proj(x)), sh(self.k_proj(x)), sh(self.v_proj(x))
        cos, sin = rope_tables(L, self.hd)
        q = q * cos + rotate_half(q) * sin
        k = k * cos + rotate_half(k) * sin
        att = (q @ k.transpose(-1, -2)) * self.hd ** -0.5
        att = att.masked_fill(~torch.ones(L, L, dtype=torch.bool).tril(), float("-inf")).softmax(-1)
        return self.o_proj((att @ v).transpose(1, 2).reshape(B, L, D))


class LlamaMLP(nn.Module):
    def __init__(self, d, m):
        super().__init__()
        self.gate_proj = nn.Linear(d, m, bias=False)
        self.down_proj = nn.Linear(m, d, bias=False)
        self.up_proj = nn.Linear(d, m, bias=False)

    def forward(self, x):
        return self.down_proj(F.silu(self.gate_proj(x)) * self.up_proj(x))


class LlamaDecoderLayer(nn.Module):
    def __init__(self, d, heads, m, eps):
        super().__init__()
        self.self_attn = LlamaAttention(d, heads)
        self.mlp = LlamaMLP(d, m)
        self.input_layernorm = RMSNorm(d, eps)
        self.post_attention_layernorm = RMSNorm(d, eps)

    def forward(self, x):
        x = x + self.self_attn(self.input_layernorm(x))
        return x + self.mlp(self.post_attention_layernorm(x))


class _LlamaModel(nn.Module):
    def __init__(self, vocab, d, layers, heads, m, eps):
        super().__init__()
        self.embed_tokens = nn.Embedding(vocab, d)
        self.layers = nn.ModuleList([LlamaDecoderLayer(d, heads, m, eps) for _ in range(layers)])
        self.norm = RMSNorm(d, eps)


class LlamaForCausalLM(nn.Module):
    def __init__(self, vocab, d, layers, heads, m, eps=1e-6):
        super().__init__()
        self.model = _LlamaModel(vocab, d, layers, heads, m, eps)
        self.lm_head = nn.Linear(d, vocab, bias=False)

    def forward(self, input_ids, labels=None):
        x = self.model.embed_tokens(input_ids)
        for l in self.model.layers:
            x = l(x)
        logits = self.lm_head(self.model.norm(x))
        loss = None
        if labels is not None:
            loss = F.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), labels[:, 1:].reshape(-1))
        return loss, logits
