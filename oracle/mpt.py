"""ORACLE (test infrastructure only): the MPT causal-LM tower behind OpenFlamingo-9B (UniMP/mmrec.py:515-524,
``lang_encoder_path="anas-awadalla/mpt-7b"``), fp32 CPU PyTorch.

The arithmetic lives in third-party ``transformers`` (requirements.txt:26 ``transformers>=4.29.0``; 5.15 installed here), not
under /root/reference: restated from its published modelling file (models/mpt/modeling_mpt.py: MptBlock, MptAttention with
``build_mpt_alibi_tensor``, MptMLP) -- bias-free LayerNorm and Linear layers, fused ``Wqkv`` chunked q | k | v, scores
``q k^T / sqrt(hd) + slope_h * (j - (L - 1))`` under a causal + key-padding mask, GELU(erf) MLP with expansion 4, tied head.
Pinned against the installed transformers' ``MptForCausalLM`` (tests/golden/mpt_tiny.npz, oracle/make_golden.py).

``MosaicGPT`` (the "3b" / "3b-instruct" towers, mmrec.py:475-494: ``anas-awadalla/mpt-1b-redpajama-200b[-dolly]``) is the same
block with LayerNorm over the full d_model vectors of q and of k before the head split (mosaic_gpt ``attn_qk_ln``; the later
llm-foundry MPT's ``qk_ln``) under mosaic_gpt's module names.  PARITY UNPINNED for that one option: the model's custom
``mosaic_gpt.py`` lives in its Hugging Face repo (absent here, no version pinned by the reference) and the installed
transformers' MPT carries ``qk_ln`` in its config but does not implement it.  Anchors: with q_ln = k_ln = identity-free path
switched off (``attn_qk_ln=False``) the tower IS the pinned MPT under renamed modules (tests/test_oracle_kat.py).
"""
import math
import torch
import torch.nn as nn
import torch.nn.functional as F

from .lm import LMOutput, _ResizeMixin, _mask_bias, hf_causal_lm_loss
from .numerics import st


class MPTConfig:
    def __init__(self, vocab_size=50432, d_model=4096, n_layers=32, n_heads=32, expansion_ratio=4, max_seq_len=2048,
                 layer_norm_epsilon=1e-5, alibi_bias_max=8):
        self.__dict__.update({k: v for k, v in locals().items() if k != "self"})
        self.model_type, self.hidden_size = "mpt", d_model


def alibi_slopes(n_heads, alibi_bias_max=8):
    n2 = 2 ** math.ceil(math.log2(n_heads))
    base = torch.arange(1, n2 + 1, dtype=torch.float32) * (alibi_bias_max / n2)
    slopes = 1.0 / torch.pow(2, base)
    if n2 != n_heads:
        slopes = torch.cat([slopes[1::2], slopes[::2]])[:n_heads]
    return slopes


class _LN(nn.Module):
    def __init__(self, d, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.eps = eps

    def forward(self, x):
        return F.layer_norm(x, (x.shape[-1],), self.weight, None, self.eps)


class _Attn(nn.Module):
    def __init__(self, c, qk_ln=False):
        super().__init__()
        self.nh, self.hd = c.n_heads, c.d_model // c.n_heads
        self.Wqkv = nn.Linear(c.d_model, 3 * c.d_model, bias=False)
        if qk_ln:
            self.q_ln, self.k_ln = _LN(c.d_model, c.layer_norm_epsilon), _LN(c.d_model, c.layer_norm_epsilon)
        self.out_proj = nn.Linear(c.d_model, c.d_model, bias=False)

    def forward(self, x, bias):
        B, L, D = x.shape
        q, k, v = self.Wqkv(x).chunk(3, dim=2)
        if hasattr(self, "q_ln"):              # over the whole d_model vector, before the heads are split
            q, k = self.q_ln(st("gemm", q)), self.k_ln(st("gemm", k))
        q, k, v = (st("gemm", t).reshape(B, L, self.nh, self.hd).transpose(1, 2) for t in (q, k, v))
        s = q @ k.transpose(-1, -2) / math.sqrt(self.hd) + bias
        p = st("attn_p", torch.softmax(s.float(), -1).to(v.dtype))
        return self.out_proj(st("attn_o", (p @ v).transpose(1, 2).reshape(B, L, D)))


class _FFN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.up_proj = nn.Linear(c.d_model, c.expansion_ratio * c.d_model, bias=False)
        self.down_proj = nn.Linear(c.expansion_ratio * c.d_model, c.d_model, bias=False)

    def forward(self, x):
        return self.down_proj(st("act", F.gelu(self.up_proj(x))))


class MptBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.norm_1, self.attn = _LN(c.d_model, c.layer_norm_epsilon), _Attn(c)
        self.norm_2, self.ffn = _LN(c.d_model, c.layer_norm_epsilon), _FFN(c)

    def forward(self, x, attention_mask=None, **kw):
        x = st("res", x + self.attn(st("ln", self.norm_1(x)), attention_mask))
        return st("res", x + self.ffn(st("ln", self.norm_2(x))))


class _Body(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.wte = nn.Embedding(c.vocab_size, c.d_model)
        self.blocks = nn.ModuleList([MptBlock(c) for _ in range(c.n_layers)])
        self.norm_f = _LN(c.d_model, c.layer_norm_epsilon)


class MptForCausalLM(nn.Module, _ResizeMixin):
    decoder_layers_attr = "transformer.blocks"
    tied = True

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.transformer = _Body(config)
        self.lm_head = nn.Linear(config.d_model, config.vocab_size, bias=False)
        self.lm_head.weight = self.transformer.wte.weight

    def get_input_embeddings(self): return self.transformer.wte
    def set_input_embeddings(self, m): self.transformer.wte = m
    def get_output_embeddings(self): return self.lm_head
    def set_output_embeddings(self, m): self.lm_head = m

    def forward(self, input_ids, attention_mask=None, labels=None, **kw):
        B, L = input_ids.shape
        x = self.transformer.wte(input_ids)
        c = self.config
        alibi = alibi_slopes(c.n_heads, c.alibi_bias_max)[None, :, None, None] * torch.arange(1 - L, 1, dtype=torch.float32)[None, None, None, :]
        bias = _mask_bias(attention_mask, L, x.dtype) + alibi
        for blk in self.transformer.blocks:
            x = blk(x, attention_mask=bias)
        logits = st("logits", self.lm_head(st("ln", self.transformer.norm_f(x))))
        loss = hf_causal_lm_loss(logits, labels) if labels is not None else None
        return LMOutput(loss, logits)


# --------------------------------------------------------------------------- MPT-1B (mosaic_gpt): QK-LayerNorm, other module names
class MosaicGPTConfig(MPTConfig):
    def __init__(self, vocab_size=50432, d_model=2048, n_layers=24, n_heads=16, mlp_ratio=4, max_seq_len=2048,
                 layer_norm_epsilon=1e-5, alibi_bias_max=8, attn_qk_ln=True):
        super().__init__(vocab_size, d_model, n_layers, n_heads, mlp_ratio, max_seq_len, layer_norm_epsilon, alibi_bias_max)
        self.model_type, self.mlp_ratio, self.attn_qk_ln = "mosaic_gpt", mlp_ratio, attn_qk_ln


class _MosaicMLP(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.mlp_up = nn.Linear(c.d_model, c.mlp_ratio * c.d_model, bias=False)
        self.mlp_down = nn.Linear(c.mlp_ratio * c.d_model, c.d_model, bias=False)

    def forward(self, x):
        return self.mlp_down(st("act", F.gelu(self.mlp_up(x))))


class MosaicGPTBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.ln_1, self.attn = _LN(c.d_model, c.layer_norm_epsilon), _Attn(c, c.attn_qk_ln)
        self.ln_2, self.mlp = _LN(c.d_model, c.layer_norm_epsilon), _MosaicMLP(c)

    def forward(self, x, attention_mask=None, **kw):
        x = st("res", x + self.attn(st("ln", self.ln_1(x)), attention_mask))
        return st("res", x + self.mlp(st("ln", self.ln_2(x))))


class _MosaicBody(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.wte = nn.Embedding(c.vocab_size, c.d_model)
        self.blocks = nn.ModuleList([MosaicGPTBlock(c) for _ in range(c.n_layers)])
        self.ln_f = _LN(c.d_model, c.layer_norm_epsilon)

    @property
    def norm_f(self):
        return self.ln_f


class MosaicGPT(MptForCausalLM):
    def __init__(self, config):
        nn.Module.__init__(self)
        self.config = config
        self.transformer = _MosaicBody(config)
        self.lm_head = nn.Linear(config.d_model, config.vocab_size, bias=False)
        self.lm_head.weight = self.transformer.wte.weight
