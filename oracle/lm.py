"""Oracle: causal-LM towers behind Flamingo (fp32, CPU).  TEST INFRASTRUCTURE ONLY.

GPT-NeoX (RedPajama-INCITE-3B = UniMP's ``4b-instruct``, UniMP/mmrec.py:505-514) and OPT
(``configs[0]`` plumbing tower).  The towers are third-party code in the reference
(``transformers>=4.29.0``, requirements.txt:26); the arithmetic restated here follows the
installed modelling files
  transformers/models/gpt_neox/modeling_gpt_neox.py:107-283 (RoPE half-split, per-head
      interleaved QKV, sequential/parallel residual, GELU MLP)
  transformers/models/opt/modeling_opt.py (learned positions with +2 offset from the mask
      cumsum, pre-LN, ReLU MLP, q scaled by hd^-0.5, tied head)
and keeps the 4.29-era parameter names (``gpt_neox.embed_in``, ``embed_out``,
``model.decoder.layers`` ...) that OpenFlamingo checkpoints use.
"""
import math
import torch
import torch.nn as nn
import torch.nn.functional as F

from .numerics import st


class LMOutput:
    """Minimal stand-in for HF ``CausalLMOutputWithPast``: ``out[0]`` is the loss when
    labels were given (else logits), ``out["logits"]`` the logits (UniMP/mmrec.py:182,190)."""

    def __init__(self, loss, logits):
        self.loss, self.logits = loss, logits

    def __getitem__(self, k):
        if isinstance(k, str):
            return getattr(self, k)
        fields = [f for f in (self.loss, self.logits) if f is not None]
        return fields[k]


def hf_causal_lm_loss(logits, labels):
    """HF's internal shifted mean CE (logged only: UniMP/mmrec.py:182,292)."""
    return F.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]).float(),
                           labels[:, 1:].reshape(-1), ignore_index=-100)


def _mask_bias(attention_mask, L, dtype):
    """causal + key-padding additive bias (B,1,L,L)."""
    causal = torch.ones(L, L, dtype=torch.bool).tril()
    ok = causal[None, None]
    if attention_mask is not None:
        ok = ok & attention_mask[:, None, None, :].bool()
    return torch.zeros(ok.shape, dtype=dtype).masked_fill(~ok, torch.finfo(dtype).min)


# --------------------------------------------------------------------------- GPT-NeoX
class NeoXConfig:
    model_type = "gpt_neox"

    def __init__(self, vocab_size=50432, hidden_size=2560, num_hidden_layers=32,
                 num_attention_heads=32, intermediate_size=10240, rotary_pct=1.0,
                 rotary_emb_base=10000.0, layer_norm_eps=1e-5, use_parallel_residual=False,
                 max_position_embeddings=2048):
        self.__dict__.update(locals())
        del self.__dict__["self"]


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), -1)


def neox_rope_tables(L, rot, base):
    inv = 1.0 / (base ** (torch.arange(0, rot, 2, dtype=torch.float32) / rot))
    fr = torch.arange(L, dtype=torch.float32)[:, None] * inv[None]
    emb = torch.cat((fr, fr), -1)
    return emb.cos(), emb.sin()


class NeoXAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.nh = c.num_attention_heads
        self.hd = c.hidden_size // self.nh
        self.rot = int(self.hd * c.rotary_pct)
        self.base = c.rotary_emb_base
        self.query_key_value = nn.Linear(c.hidden_size, 3 * c.hidden_size)
        self.dense = nn.Linear(c.hidden_size, c.hidden_size)

    def forward(self, x, bias):
        B, L, H = x.shape
        qkv = self.query_key_value(x).view(B, L, self.nh, 3 * self.hd).transpose(1, 2)
        q, k, v = qkv.chunk(3, -1)                      # per-head interleave [h][q,k,v]
        cos, sin = neox_rope_tables(L, self.rot, self.base)
        qr, qp = q[..., :self.rot], q[..., self.rot:]
        kr, kp = k[..., :self.rot], k[..., self.rot:]
        q = st("gemm", torch.cat((qr * cos + rotate_half(qr) * sin, qp), -1))
        k = st("gemm", torch.cat((kr * cos + rotate_half(kr) * sin, kp), -1))
        v = st("gemm", v)
        att = (q @ k.transpose(-1, -2)) * self.hd ** -0.5 + bias
        att = st("attn_p", att.softmax(-1))
        o = st("attn_o", (att @ v).transpose(1, 2).reshape(B, L, H))
        return self.dense(o)


class NeoXMLP(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense_h_to_4h = nn.Linear(c.hidden_size, c.intermediate_size)
        self.dense_4h_to_h = nn.Linear(c.intermediate_size, c.hidden_size)

    def forward(self, x):
        return self.dense_4h_to_h(st("act", F.gelu(self.dense_h_to_4h(x))))


class NeoXLayer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.par = c.use_parallel_residual
        self.input_layernorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.post_attention_layernorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.attention = NeoXAttention(c)
        self.mlp = NeoXMLP(c)

    def forward(self, x, attention_mask=None, **kw):
        a = self.attention(st("ln", self.input_layernorm(x)), attention_mask)
        if self.par:
            return st("res", self.mlp(st("ln", self.post_attention_layernorm(x))) + st("res", a + x))
        a = st("res", a + x)
        return st("res", self.mlp(st("ln", self.post_attention_layernorm(a))) + a)


class _NeoXBody(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.embed_in = nn.Embedding(c.vocab_size, c.hidden_size)
        self.layers = nn.ModuleList([NeoXLayer(c) for _ in range(c.num_hidden_layers)])
        self.final_layer_norm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class _ResizeMixin:
    def resize_token_embeddings(self, n):
        """HF semantics (4.29): new Embedding / Linear modules, old rows copied, new rows
        ~N(0, 0.02); fresh modules => requires_grad=True on both."""
        old = self.get_input_embeddings()
        if n == old.weight.shape[0]:
            return old
        new = nn.Embedding(n, old.weight.shape[1])
        new.weight.data.normal_(0, 0.02)
        k = min(n, old.weight.shape[0])
        new.weight.data[:k] = old.weight.data[:k]
        self.set_input_embeddings(new)
        head = self.get_output_embeddings()
        if self.tied:
            head.weight = new.weight
        else:
            nh = nn.Linear(head.weight.shape[1], n, bias=False)
            nh.weight.data.normal_(0, 0.02)
            nh.weight.data[:k] = head.weight.data[:k]
            self.set_output_embeddings(nh)
        self.config.vocab_size = n
        return new


class GPTNeoXForCausalLM(nn.Module, _ResizeMixin):
    decoder_layers_attr = "gpt_neox.layers"
    tied = False

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.gpt_neox = _NeoXBody(config)
        self.embed_out = nn.Linear(config.hidden_size, config.vocab_size, bias=False)

    def get_input_embeddings(self): return self.gpt_neox.embed_in
    def set_input_embeddings(self, m): self.gpt_neox.embed_in = m
    def get_output_embeddings(self): return self.embed_out
    def set_output_embeddings(self, m): self.embed_out = m

    def forward(self, input_ids, attention_mask=None, labels=None, **kw):
        x = self.gpt_neox.embed_in(input_ids)
        bias = _mask_bias(attention_mask, input_ids.shape[1], x.dtype)
        for layer in self.gpt_neox.layers:
            x = layer(x, attention_mask=bias)
        logits = st("logits", self.embed_out(st("ln", self.gpt_neox.final_layer_norm(x))))
        loss = hf_causal_lm_loss(logits, labels) if labels is not None else None
        return LMOutput(loss, logits)


# --------------------------------------------------------------------------- OPT
class OPTConfig:
    model_type = "opt"

    def __init__(self, vocab_size=50272, hidden_size=768, num_hidden_layers=12,
                 num_attention_heads=12, ffn_dim=3072, max_position_embeddings=2048):
        self.__dict__.update(locals())
        del self.__dict__["self"]


class OPTAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.nh, self.hd = c.num_attention_heads, c.hidden_size // c.num_attention_heads
        d = c.hidden_size
        self.k_proj, self.v_proj = nn.Linear(d, d), nn.Linear(d, d)
        self.q_proj, self.out_proj = nn.Linear(d, d), nn.Linear(d, d)

    def forward(self, x, bias):
        B, L, D = x.shape
        sh = lambda t: t.view(B, L, self.nh, self.hd).transpose(1, 2)
        q = sh(self.q_proj(x) * self.hd ** -0.5)
        k, v = sh(self.k_proj(x)), sh(self.v_proj(x))
        att = (q @ k.transpose(-1, -2) + bias).softmax(-1)
        return self.out_proj((att @ v).transpose(1, 2).reshape(B, L, D))


class OPTDecoderLayer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.self_attn = OPTAttention(c)
        self.self_attn_layer_norm = nn.LayerNorm(c.hidden_size)
        self.fc1 = nn.Linear(c.hidden_size, c.ffn_dim)
        self.fc2 = nn.Linear(c.ffn_dim, c.hidden_size)
        self.final_layer_norm = nn.LayerNorm(c.hidden_size)

    def forward(self, x, attention_mask=None, **kw):
        x = x + self.self_attn(self.self_attn_layer_norm(x), attention_mask)
        return x + self.fc2(F.relu(self.fc1(self.final_layer_norm(x))))


class _OPTDecoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.embed_tokens = nn.Embedding(c.vocab_size, c.hidden_size)
        self.embed_positions = nn.Embedding(c.max_position_embeddings + 2, c.hidden_size)
        self.final_layer_norm = nn.LayerNorm(c.hidden_size)
        self.layers = nn.ModuleList([OPTDecoderLayer(c) for _ in range(c.num_hidden_layers)])


class _OPTModel(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.decoder = _OPTDecoder(c)


class OPTForCausalLM(nn.Module, _ResizeMixin):
    decoder_layers_attr = "model.decoder.layers"
    tied = True

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.model = _OPTModel(config)
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.lm_head.weight = self.model.decoder.embed_tokens.weight

    def get_input_embeddings(self): return self.model.decoder.embed_tokens
    def set_input_embeddings(self, m): self.model.decoder.embed_tokens = m
    def get_output_embeddings(self): return self.lm_head
    def set_output_embeddings(self, m): self.lm_head = m

    def forward(self, input_ids, attention_mask=None, labels=None, **kw):
        d = self.model.decoder
        B, L = input_ids.shape
        am = attention_mask if attention_mask is not None else torch.ones(B, L, dtype=torch.long)
        pos = (torch.cumsum(am, 1) * am).long() - 1 + 2           # OPTLearnedPositionalEmbedding
        x = d.embed_tokens(input_ids) + d.embed_positions(pos)
        bias = _mask_bias(attention_mask, L, x.dtype)
        for layer in d.layers:
            x = layer(x, attention_mask=bias)
        logits = self.lm_head(d.final_layer_norm(x))
        loss = hf_causal_lm_loss(logits, labels) if labels is not None else None
        return LMOutput(loss, logits)
