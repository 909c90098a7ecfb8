"""Causal-LM towers behind Flamingo, on the HIP kernels.

GPT-NeoX (RedPajama-INCITE-3B = UniMP ``4b-instruct``, UniMP/mmrec.py:505-514) and OPT (plumbing config).
Module tree / parameter names are those of transformers 4.29 (``gpt_neox.embed_in``, ``embed_out``,
``model.decoder.layers`` ...), which OpenFlamingo's mixin and checkpoints address;
arithmetic follows transformers' gpt_neox / opt modelling files (RoPE half-split on the first rotary_ndims,
per-head interleaved QKV, sequential or parallel residual, GELU / ReLU MLP, OPT positions = cumsum(mask)+1).
The towers own their layers (they do not wrap HF modules), so the HF-version drift noted in SURVEY.md §7 is moot.
"""
import math
import torch
import torch.nn as nn

from . import functional as F_
from . import ops

bf16 = torch.bfloat16


class LMOutput:
    """``out[0]`` = loss when labels were passed (else logits); ``out["logits"]`` (UniMP/mmrec.py:182,190)."""

    def __init__(self, loss, logits, stats=None, past_key_values=None, hidden_rows=None):
        self.loss, self.logits, self.stats, self.past_key_values = loss, logits, stats, past_key_values
        self.hidden_rows = hidden_rows          # head_rows path: final hidden states of the requested rows, no logits

    def __getitem__(self, k):
        if isinstance(k, str):
            return getattr(self, k)
        return [f for f in (self.loss, self.logits) if f is not None][k]


class _Cfg:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def NeoXConfig(vocab_size=50432, hidden_size=2560, num_hidden_layers=32, num_attention_heads=32,
               intermediate_size=10240, rotary_pct=1.0, rotary_emb_base=10000.0, layer_norm_eps=1e-5,
               use_parallel_residual=False, max_position_embeddings=2048):
    return _Cfg(model_type="gpt_neox", **{k: v for k, v in locals().items()})


def OPTConfig(vocab_size=50272, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, ffn_dim=3072,
              max_position_embeddings=2048):
    return _Cfg(model_type="opt", **{k: v for k, v in locals().items()})


LM_CONFIGS = {
    # name fragments -> config factory (no network: weights are random-init or loaded from a local state_dict)
    "RedPajama-INCITE-Instruct-3B-v1": lambda: NeoXConfig(),
    "RedPajama-INCITE-Base-3B-v1": lambda: NeoXConfig(),
    "pythia-160m": lambda: NeoXConfig(vocab_size=50304, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                                      intermediate_size=3072, rotary_pct=0.25, use_parallel_residual=True),
    "opt-125m": lambda: OPTConfig(),
    "opt-1.3b": lambda: OPTConfig(hidden_size=2048, num_hidden_layers=24, num_attention_heads=32, ffn_dim=8192),
    "mpt-7b": lambda: MPTConfig(),                                                      # OpenFlamingo-9B (mmrec.py:515-524)
    # the "3b" / "3b-instruct" options of mmrec.py:475-494: MosaicML's MPT-1B (mosaic_gpt), the QK-LayerNorm variant
    "mpt-1b-redpajama-200b-dolly": lambda: MosaicGPTConfig(),
    "mpt-1b-redpajama-200b": lambda: MosaicGPTConfig(),
}


def _pad8(n):
    return (n + 7) // 8 * 8


class _TowerBase(nn.Module):
    """shared: resize_token_embeddings (HF semantics), kv_len from the attention mask, head + loss."""
    tied = False
    packed = None              # packed token order for THIS tower: True / False (Trainer(packed=...)), None = the UNIMP_PACKED default
    supports_packed = True     # False: the tower's blocks have no packed-row form (Llama; MosaicGPT with QK-LayerNorm)

    def _pack(self, attention_mask, cache):
        """the functional.Pack of a training forward in packed token order (Trainer(packed=True)), or None.  The flag lives on the
        tower (a second Trainer does not change the first one's behaviour); a tower without a packed-row form keeps the padded
        path under the environment default and is refused by Trainer(packed=True) up front."""
        want = F_.PACKED if self.packed is None else self.packed
        if not (want and self.supports_packed and cache is None and attention_mask is not None and self.training):
            return None
        pack = F_.Pack(attention_mask)                     # packed token order: the tower never computes the <PAD> rows
        return pack if pack.useful else None

    def _run_packed(self, pack, input_ids, emb_weight, layers, pos_emb=None, **layer_kw):
        """embedding of the valid tokens -> the decoder layers on [1, M, H] packed rows -> padded [B, L, H] (zeros at <PAD>).
        pos_emb = (position ids [B, L], table): learned positional embeddings added to the token embeddings (OPT)."""
        B, L = input_ids.shape
        ids = input_ids.reshape(-1)[pack.idx_long].view(1, pack.M)
        if pos_emb is not None:
            x = F_.embedding(ids, emb_weight, pos_emb[0].reshape(-1)[pack.idx_long].view(1, pack.M), pos_emb[1])
        else:
            x = F_.embedding(ids, emb_weight)
        F_.PACK = pack
        try:
            for layer in layers:
                x = layer(x, **layer_kw)
        finally:
            F_.PACK = None
        return F_.UnpackRowsFn.apply(x.view(pack.M, -1), pack).view(B, L, -1)

    def resize_token_embeddings(self, n, preserve_requires_grad=False):
        """HF ``resize_token_embeddings`` (mmrec.py:595).  With the reference's pinned transformers (4.29,
        requirements.txt:26) the new nn.Embedding -- and, for an untied tower, the new nn.Linear head -- are fresh modules
        whose weights have ``requires_grad=True``, whatever the factory froze before: after mmrec.py:595 the 190 M-parameter
        output head of the 4b-instruct tower is therefore TRAINABLE (the 1.34 B trainable parameters of
        bench.py's cfg2 include it) and ``freeze_lm_embeddings=True`` is undone.  That is the default here too.
        ``preserve_requires_grad=True`` keeps the old modules' flags instead (what newer transformers releases do)."""
        old = self.get_input_embeddings()
        if n == old.weight.shape[0]:
            return old
        new = nn.Embedding(n, old.weight.shape[1], device=old.weight.device, dtype=old.weight.dtype)
        new.weight.data.normal_(0, 0.02)
        k = min(n, old.weight.shape[0])
        new.weight.data[:k] = old.weight.data[:k]
        if preserve_requires_grad:
            new.weight.requires_grad_(old.weight.requires_grad)
        self.set_input_embeddings(new)
        head = self.get_output_embeddings()
        if self.tied:
            head.weight = new.weight
        else:
            nh = nn.Linear(head.weight.shape[1], n, bias=False, device=head.weight.device, dtype=head.weight.dtype)
            nh.weight.data.normal_(0, 0.02)
            nh.weight.data[:k] = head.weight.data[:k]
            if preserve_requires_grad:
                nh.weight.requires_grad_(head.weight.requires_grad)
            self.set_output_embeddings(nh)
        self.config.vocab_size = n
        return new

    def _get_decoder_layers(self):
        o = self
        for part in self.decoder_layers_attr.split("."):
            o = getattr(o, part)
        return o

    def is_conditioned(self):
        return all(l.is_conditioned() for l in self._get_decoder_layers())

    def clear_conditioned_layers(self):
        for l in self._get_decoder_layers():
            l.condition_vis_x(None)
            l.condition_media_locations(None)
            l.condition_media_time(None)
            l.condition_use_cached_media(None)

    @staticmethod
    def _kv_len(attention_mask):
        if attention_mask is None:
            return None
        return attention_mask.sum(1).to(torch.int32)          # right padding (collate_rec.py:38-74)

    def _decode_state(self, past_key_values, use_cache, labels, attention_mask):
        """(cache, pos0) for a forward with ``use_cache`` / ``past_key_values`` (inference, unpadded rows), else (None, 0)."""
        if past_key_values is None and not use_cache:
            return None, 0
        if labels is not None or torch.is_grad_enabled():
            raise NotImplementedError("the KV cache is a decoding path: call under torch.no_grad() and without labels")
        cache = past_key_values if past_key_values is not None else F_.DecodeCache(len(self._get_decoder_layers()))
        if attention_mask is not None and not bool(attention_mask.all()):
            # right-padded prompts are fine for the PREFILL (keys beyond a row's length are masked by kv_len and the decode
            # steps then run with per-row positions: decode.DecodeSession); a host-position step needs equal lengths
            if cache.len != 0:
                raise NotImplementedError("past_key_values with padded rows: use decode.DecodeSession (per-row positions)")
            cache.row_len = attention_mask.sum(1).to(torch.int32)
        return cache, cache.len

    @staticmethod
    def _step_rope(cache, rope):
        """static decode step: the table row of the device-side position, gathered once per forward for all layers."""
        if cache is None or cache.step is None:
            return rope
        return rope[0].index_select(0, cache.step.pos_idx), rope[1].index_select(0, cache.step.pos_idx), rope[2]

    def _head(self, h, labels, last_only=False, head_rows=None, last_index=None):
        if last_index is not None:          # padded prompts: every row's own last valid position
            h = h[torch.arange(h.shape[0], device=h.device), last_index].unsqueeze(1)
            last_only = False
        if isinstance(head_rows, str):     # "hidden": the caller applies head + loss as one fused node (train.py, functional.DenseHeadLossFn)
            return LMOutput(None, None, hidden_rows=h)
        if head_rows is not None:          # the caller applies the head itself on these flattened (b*L + j) rows (train.py)
            return LMOutput(None, None, hidden_rows=h.reshape(-1, h.shape[-1]).index_select(0, head_rows))
        w = self.get_output_embeddings().weight
        V = w.shape[0]
        if last_only:                      # decoding: score only the final position (generate.py)
            h = h[:, -1:].contiguous()
        logits = F_.linear(h, w, None, _pad8(V) if V % 8 else None)
        loss = stats = None
        if labels is not None:
            ones = torch.ones(h.shape[0], dtype=torch.float32, device=h.device)
            loss, stats = F_.focal_ce(logits, labels.to(h.device), ones, 0.0, False)   # HF mean CE (mmrec.py:182)
        return LMOutput(loss, logits, stats)


# --------------------------------------------------------------------------- GPT-NeoX
class _NeoXAttnParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.query_key_value = nn.Linear(c.hidden_size, 3 * c.hidden_size)
        self.dense = nn.Linear(c.hidden_size, c.hidden_size)


class _NeoXMLPParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense_h_to_4h = nn.Linear(c.hidden_size, c.intermediate_size)
        self.dense_4h_to_h = nn.Linear(c.intermediate_size, c.hidden_size)


class GPTNeoXLayer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.c = c
        self.input_layernorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.post_attention_layernorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.attention = _NeoXAttnParams(c)
        self.mlp = _NeoXMLPParams(c)

    def forward(self, x, attention_mask=None, rope=None, cache=None, pos0=0, **kw):
        c, a, m = self.c, self.attention, self.mlp
        l1, l2 = self.input_layernorm, self.post_attention_layernorm
        if cache is not None:
            att = F_.self_attn_block_cached(x, l1.weight, l1.bias, a.query_key_value.weight, a.query_key_value.bias,
                                            a.dense.weight, a.dense.bias, c.num_attention_heads, cache, pos0, rope=rope,
                                            interleaved=True, eps=l1.eps)
        else:
            att = F_.self_attn_block(x, l1.weight, l1.bias, a.query_key_value.weight, a.query_key_value.bias, a.dense.weight,
                                     a.dense.bias, c.num_attention_heads, rope=rope, kv_len=attention_mask, interleaved=True,
                                     causal=True, eps=l1.eps)
        if c.use_parallel_residual:      # x + attn(ln1(x)) + mlp(ln2(x))
            return F_.mlp_block(x, l2.weight, l2.bias, m.dense_h_to_4h.weight, m.dense_h_to_4h.bias, m.dense_4h_to_h.weight,
                                m.dense_4h_to_h.bias, "gelu", res=att, eps=l2.eps)
        return F_.mlp_block(att, l2.weight, l2.bias, m.dense_h_to_4h.weight, m.dense_h_to_4h.bias, m.dense_4h_to_h.weight,
                            m.dense_4h_to_h.bias, "gelu", eps=l2.eps)


class _NeoXBody(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.embed_in = nn.Embedding(c.vocab_size, c.hidden_size)
        self.layers = nn.ModuleList([GPTNeoXLayer(c) for _ in range(c.num_hidden_layers)])
        self.final_layer_norm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class GPTNeoXForCausalLM(_TowerBase):
    decoder_layers_attr = "gpt_neox.layers"

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.gpt_neox = _NeoXBody(config)
        self.embed_out = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self._rope = None

    def get_input_embeddings(self): return self.gpt_neox.embed_in
    def set_input_embeddings(self, m): self.gpt_neox.embed_in = m
    def get_output_embeddings(self): return self.embed_out
    def set_output_embeddings(self, m): self.embed_out = m

    def _rope_tables(self, L, device):
        c = self.config
        rot = int((c.hidden_size // c.num_attention_heads) * c.rotary_pct)
        if self._rope is None or self._rope[0].shape[0] < L or self._rope[0].device != device:
            n = max(L, 512)
            inv = 1.0 / (c.rotary_emb_base ** (torch.arange(0, rot, 2, dtype=torch.float32) / rot))
            fr = torch.arange(n, dtype=torch.float32)[:, None] * inv[None]
            self._rope = (fr.cos().contiguous().to(device), fr.sin().contiguous().to(device), rot, float(c.rotary_emb_base))
        return self._rope

    def forward(self, input_ids, attention_mask=None, labels=None, past_key_values=None, use_cache=False, **kw):
        B, L = input_ids.shape
        cache, pos0 = self._decode_state(past_key_values, use_cache, labels, attention_mask)
        kv_len = self._kv_len(attention_mask)
        pack = self._pack(attention_mask, cache)
        if pack is not None:
            rope = self._rope_tables(L, input_ids.device)
            x = self._run_packed(pack, input_ids, self.gpt_neox.embed_in.weight, self.gpt_neox.layers, attention_mask=kv_len, rope=rope, cache=None, pos0=0)
        else:
            x = F_.embedding(input_ids, self.gpt_neox.embed_in.weight)
            rope = self._step_rope(cache, self._rope_tables(max(pos0 + L, cache.kv.shape[3] if cache and cache.kv is not None else 0), x.device))
            for i, layer in enumerate(self.gpt_neox.layers):
                x = layer(x, attention_mask=kv_len, rope=rope, cache=cache.layers[i] if cache else None, pos0=pos0)
        f = self.gpt_neox.final_layer_norm
        if (cache is not None and cache.step is not None and labels is None and kw.get("head_rows") is None and kw.get("last_index") is None
                and not torch.is_grad_enabled() and F_._ln_fusable(B * L, x.shape[-1], False)):
            # a cached decode step: the final LayerNorm runs inside the head's weight-streaming GEMM (one launch for 379 MB of weights)
            w = self.get_output_embeddings().weight
            V = w.shape[0]
            out = LMOutput(None, F_.linear_ln(x, f.weight, f.bias, f.eps, w, _pad8(V) if V % 8 else None))
        else:
            h = F_.layer_norm(x, f.weight, f.bias, f.eps)
            out = self._head(h, labels, kw.get("logits_last_only", False), kw.get("head_rows"), kw.get("last_index"))
        if cache is not None:
            if cache.step is None:
                cache.len = pos0 + L
            out.past_key_values = cache
        return out


# --------------------------------------------------------------------------- OPT
class _OPTAttnParams(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.k_proj, self.v_proj = nn.Linear(d, d), nn.Linear(d, d)
        self.q_proj, self.out_proj = nn.Linear(d, d), nn.Linear(d, d)
        self._fused = None

    def fused_qkv(self):
        """blocked [q;k;v] copy of the three frozen projections (rebuilt when any of them changes)."""
        ps = [self.q_proj.weight, self.k_proj.weight, self.v_proj.weight, self.q_proj.bias, self.k_proj.bias, self.v_proj.bias]
        if torch.is_grad_enabled() and any(p.requires_grad for p in ps):
            raise NotImplementedError("training OPT q/k/v projections is not supported (the LM tower is frozen in UniMP)")
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if self._fused is None or self._fused[0] != key:
            self._fused = (key, torch.cat([p.detach() for p in ps[:3]]).contiguous(), torch.cat([p.detach() for p in ps[3:]]).contiguous())
        return self._fused[1], self._fused[2]


class OPTDecoderLayer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.c = c
        self.self_attn = _OPTAttnParams(c.hidden_size)
        self.self_attn_layer_norm = nn.LayerNorm(c.hidden_size)
        self.fc1 = nn.Linear(c.hidden_size, c.ffn_dim)
        self.fc2 = nn.Linear(c.ffn_dim, c.hidden_size)
        self.final_layer_norm = nn.LayerNorm(c.hidden_size)

    def forward(self, x, attention_mask=None, cache=None, pos0=0, **kw):
        a, l1, l2 = self.self_attn, self.self_attn_layer_norm, self.final_layer_norm
        wqkv, bqkv = a.fused_qkv()
        if cache is not None:
            x = F_.self_attn_block_cached(x, l1.weight, l1.bias, wqkv, bqkv, a.out_proj.weight, a.out_proj.bias,
                                          self.c.num_attention_heads, cache, pos0, interleaved=False, eps=l1.eps)
        else:
            x = F_.self_attn_block(x, l1.weight, l1.bias, wqkv, bqkv, a.out_proj.weight, a.out_proj.bias,
                                   self.c.num_attention_heads, kv_len=attention_mask, interleaved=False, causal=True, eps=l1.eps)
        return F_.mlp_block(x, l2.weight, l2.bias, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, "relu", eps=l2.eps)


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


class OPTForCausalLM(_TowerBase):
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

    def forward(self, input_ids, attention_mask=None, labels=None, past_key_values=None, use_cache=False, **kw):
        d = self.model.decoder
        B, L = input_ids.shape
        cache, pos0 = self._decode_state(past_key_values, use_cache, labels, attention_mask)
        am = attention_mask if attention_mask is not None else torch.ones_like(input_ids)
        pos = (torch.cumsum(am, 1) * am).long() + 1                      # OPTLearnedPositionalEmbedding (offset 2)
        pos = pos + (cache.step.pos_idx[:, None] if cache is not None and cache.step is not None else pos0)
        kv_len = self._kv_len(attention_mask)
        pack = self._pack(attention_mask, cache)
        if pack is not None:
            x = self._run_packed(pack, input_ids, d.embed_tokens.weight, d.layers, pos_emb=(pos, d.embed_positions.weight), attention_mask=kv_len,
                                 cache=None, pos0=0)
        else:
            x = F_.embedding(input_ids, d.embed_tokens.weight, pos, d.embed_positions.weight)
            for i, layer in enumerate(d.layers):
                x = layer(x, attention_mask=kv_len, cache=cache.layers[i] if cache else None, pos0=pos0)
        f = d.final_layer_norm
        h = F_.layer_norm(x, f.weight, f.bias, f.eps)
        out = self._head(h, labels, kw.get("logits_last_only", False), kw.get("head_rows"), kw.get("last_index"))
        if cache is not None:
            if cache.step is None:
                cache.len = pos0 + L
            out.past_key_values = cache
        return out


# --------------------------------------------------------------------------- Llama (in-tree UniMP/xformers_model/llama.py)
def LlamaConfig(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                rms_norm_eps=1e-6, max_position_embeddings=2048, rope_base=10000.0):
    return _Cfg(model_type="llama", **{k: v for k, v in locals().items()})


class _Frozen:
    """cache of a fused copy of several frozen parameters (rebuilt when any of them changes)."""

    def __init__(self):
        self._c = None

    def get(self, params, what):
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            raise NotImplementedError(f"training the {what} projections is not supported (the LM tower is frozen in UniMP)")
        key = tuple((p.data_ptr(), p._version) for p in params)
        if self._c is None or self._c[0] != key:
            self._c = (key, torch.cat([p.detach() for p in params]).contiguous())
        return self._c[1]


class _LlamaAttnParams(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.q_proj = nn.Linear(d, d, bias=False)
        self.k_proj = nn.Linear(d, d, bias=False)
        self.v_proj = nn.Linear(d, d, bias=False)
        self.o_proj = nn.Linear(d, d, bias=False)


class _LlamaMLPParams(nn.Module):
    def __init__(self, d, m):
        super().__init__()
        self.gate_proj = nn.Linear(d, m, bias=False)
        self.down_proj = nn.Linear(m, d, bias=False)
        self.up_proj = nn.Linear(d, m, bias=False)


class _RMSNormParams(nn.Module):
    def __init__(self, d, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.variance_epsilon = eps


class LlamaDecoderLayer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.c = c
        self.self_attn = _LlamaAttnParams(c.hidden_size)
        self.mlp = _LlamaMLPParams(c.hidden_size, c.intermediate_size)
        self.input_layernorm = _RMSNormParams(c.hidden_size, c.rms_norm_eps)
        self.post_attention_layernorm = _RMSNormParams(c.hidden_size, c.rms_norm_eps)
        self._fqkv, self._fgu = _Frozen(), _Frozen()

    def forward(self, x, attention_mask=None, rope=None, **kw):
        a, m, c = self.self_attn, self.mlp, self.c
        wqkv = self._fqkv.get([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], "Llama q/k/v")
        wgu = self._fgu.get([m.gate_proj.weight, m.up_proj.weight], "Llama gate/up")
        # training path of the in-tree model: LowerTriangularMask only, padding masks are ignored (llama.py:287-293)
        if kw.get("cache") is not None:
            x = F_.self_attn_block_cached(x, self.input_layernorm.weight, None, wqkv, None, a.o_proj.weight, None,
                                          c.num_attention_heads, kw["cache"], kw.get("pos0", 0), rope=rope, interleaved=False,
                                          eps=c.rms_norm_eps, rms=True)
        else:
            x = F_.self_attn_block(x, self.input_layernorm.weight, None, wqkv, None, a.o_proj.weight, None,
                                   c.num_attention_heads, rope=rope, kv_len=None, interleaved=False, causal=True,
                                   eps=c.rms_norm_eps, rms=True)
        return F_.swiglu_block(x, self.post_attention_layernorm.weight, wgu, m.down_proj.weight, c.rms_norm_eps)


class _LlamaModel(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.embed_tokens = nn.Embedding(c.vocab_size, c.hidden_size)
        self.layers = nn.ModuleList([LlamaDecoderLayer(c) for _ in range(c.num_hidden_layers)])
        self.norm = _RMSNormParams(c.hidden_size, c.rms_norm_eps)


class LlamaForCausalLM(_TowerBase):
    decoder_layers_attr = "model.layers"
    supports_packed = False

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.model = _LlamaModel(config)
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self._rope = None

    def get_input_embeddings(self): return self.model.embed_tokens
    def set_input_embeddings(self, m): self.model.embed_tokens = m
    def get_output_embeddings(self): return self.lm_head
    def set_output_embeddings(self, m): self.lm_head = m

    def _rope_tables(self, L, device):
        c = self.config
        hd = c.hidden_size // c.num_attention_heads
        if self._rope is None or self._rope[0].shape[0] < L or self._rope[0].device != device:
            n = max(L, 512)
            inv = 1.0 / (c.rope_base ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
            fr = torch.arange(n, dtype=torch.float32)[:, None] * inv[None]
            self._rope = (fr.cos().contiguous().to(device), fr.sin().contiguous().to(device), hd, float(c.rope_base))
        return self._rope

    def forward(self, input_ids, attention_mask=None, labels=None, past_key_values=None, use_cache=False, **kw):
        B, L = input_ids.shape
        cache, pos0 = self._decode_state(past_key_values, use_cache, labels, None)
        x = F_.embedding(input_ids, self.model.embed_tokens.weight)
        rope = self._step_rope(cache, self._rope_tables(max(pos0 + L, cache.kv.shape[3] if cache and cache.kv is not None else 0), x.device))
        for i, layer in enumerate(self.model.layers):
            x = layer(x, attention_mask=None, rope=rope, cache=cache.layers[i] if cache else None, pos0=pos0)
        n = self.model.norm
        h = F_.layer_norm(x, n.weight, None, n.variance_epsilon, rms=True)
        out = self._head(h, labels, kw.get("logits_last_only", False), kw.get("head_rows"), kw.get("last_index"))
        if cache is not None:
            if cache.step is None:
                cache.len = pos0 + L
            out.past_key_values = cache
        return out


# --------------------------------------------------------------------------- MPT (OpenFlamingo-9B / -3B towers, mmrec.py:515-524)
def MPTConfig(vocab_size=50432, d_model=4096, n_layers=32, n_heads=32, expansion_ratio=4, max_seq_len=2048,
              layer_norm_epsilon=1e-5, alibi_bias_max=8):
    c = _Cfg(model_type="mpt", **{k: v for k, v in locals().items()})
    c.hidden_size = d_model
    return c


def mpt_alibi_slopes(n_heads, alibi_bias_max=8):
    """transformers' build_mpt_alibi_tensor: 2^-(i * bias_max / n2), interleaved when n_heads is not a power of two."""
    n2 = 2 ** math.ceil(math.log2(n_heads))
    base = torch.arange(1, n2 + 1, dtype=torch.float32) * (alibi_bias_max / n2)
    slopes = 1.0 / torch.pow(2, base)
    if n2 != n_heads:
        slopes = torch.cat([slopes[1::2], slopes[::2]])[:n_heads]
    return slopes.contiguous()


class _MPTAttnParams(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.Wqkv = nn.Linear(d, 3 * d, bias=False)
        self.out_proj = nn.Linear(d, d, bias=False)


class _MPTFFNParams(nn.Module):
    def __init__(self, d, r):
        super().__init__()
        self.up_proj = nn.Linear(d, r * d, bias=False)
        self.down_proj = nn.Linear(r * d, d, bias=False)


class _LNNoBias(nn.Module):
    def __init__(self, d, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.eps = eps


class MptBlock(nn.Module):
    """x += out_proj(causal_attn_alibi(Wqkv(norm_1(x))));  x += down_proj(gelu(up_proj(norm_2(x))))  -- no biases anywhere
    (transformers models/mpt/modeling_mpt.py MptBlock / MptAttention / MptMLP)."""

    def __init__(self, c):
        super().__init__()
        self.c = c
        self.norm_1 = _LNNoBias(c.d_model, c.layer_norm_epsilon)
        self.attn = _MPTAttnParams(c.d_model)
        self.norm_2 = _LNNoBias(c.d_model, c.layer_norm_epsilon)
        self.ffn = _MPTFFNParams(c.d_model, c.expansion_ratio)

    def forward(self, x, attention_mask=None, alibi=None, cache=None, pos0=0, **kw):
        c, a, f = self.c, self.attn, self.ffn
        if cache is not None:
            x = F_.self_attn_block_cached(x, self.norm_1.weight, None, a.Wqkv.weight, None, a.out_proj.weight, None, c.n_heads,
                                          cache, pos0, interleaved=False, eps=self.norm_1.eps, alibi=alibi)
        else:
            x = F_.self_attn_block(x, self.norm_1.weight, None, a.Wqkv.weight, None, a.out_proj.weight, None, c.n_heads,
                                   kv_len=attention_mask, interleaved=False, causal=True, eps=self.norm_1.eps, alibi=alibi)
        return F_.mlp_block(x, self.norm_2.weight, None, f.up_proj.weight, None, f.down_proj.weight, None, "gelu", eps=self.norm_2.eps)


class _MPTBody(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.wte = nn.Embedding(c.vocab_size, c.d_model)
        self.blocks = nn.ModuleList([MptBlock(c) for _ in range(c.n_layers)])
        self.norm_f = _LNNoBias(c.d_model, c.layer_norm_epsilon)


class MptForCausalLM(_TowerBase):
    decoder_layers_attr = "transformer.blocks"
    tied = True

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.transformer = _MPTBody(config)
        self.lm_head = nn.Linear(config.d_model, config.vocab_size, bias=False)
        self.lm_head.weight = self.transformer.wte.weight
        self._slopes = None

    def get_input_embeddings(self): return self.transformer.wte
    def set_input_embeddings(self, m): self.transformer.wte = m
    def get_output_embeddings(self): return self.lm_head
    def set_output_embeddings(self, m): self.lm_head = m

    def forward(self, input_ids, attention_mask=None, labels=None, past_key_values=None, use_cache=False, **kw):
        B, L = input_ids.shape
        cache, pos0 = self._decode_state(past_key_values, use_cache, labels, attention_mask)
        if self._slopes is None or self._slopes.device != input_ids.device:
            self._slopes = mpt_alibi_slopes(self.config.n_heads, self.config.alibi_bias_max).to(input_ids.device)
        kv_len = self._kv_len(attention_mask)
        pack = self._pack(attention_mask, cache)
        if pack is not None:
            x = self._run_packed(pack, input_ids, self.transformer.wte.weight, self.transformer.blocks, attention_mask=kv_len, alibi=self._slopes,
                                 cache=None, pos0=0)
        else:
            x = F_.embedding(input_ids, self.transformer.wte.weight)
            for i, blk in enumerate(self.transformer.blocks):
                x = blk(x, attention_mask=kv_len, alibi=self._slopes, cache=cache.layers[i] if cache else None, pos0=pos0)
        f = self.transformer.norm_f
        h = F_.layer_norm(x, f.weight, None, f.eps)
        out = self._head(h, labels, kw.get("logits_last_only", False), kw.get("head_rows"), kw.get("last_index"))
        if cache is not None:
            if cache.step is None:
                cache.len = pos0 + L
            out.past_key_values = cache
        return out


# --------------------------------------------------------------------------- MPT-1B / mosaic_gpt (OpenFlamingo-3B towers, mmrec.py:475-494)
def MosaicGPTConfig(vocab_size=50432, d_model=2048, n_layers=24, n_heads=16, mlp_ratio=4, max_seq_len=2048,
                    layer_norm_epsilon=1e-5, alibi_bias_max=8, attn_qk_ln=True):
    """``anas-awadalla/mpt-1b-redpajama-200b[-dolly]`` (the repo's own ``mosaic_gpt.py``, third-party: requirements.txt pins nothing
    for it; mmrec.py:475-494 names it).  Published architecture: ALiBi, no biases, LayerNorm over the FULL d_model vectors of q
    and of k before the head split (``attn_qk_ln``), GELU MLP with ratio 4, tied head.  16 heads: a power of two, where
    mosaic_gpt's ALiBi slopes 2^-(i * bias_max / n_heads) equal transformers' (``mpt_alibi_slopes``)."""
    c = _Cfg(model_type="mosaic_gpt", **{k: v for k, v in locals().items()})
    c.hidden_size = d_model
    return c


class _MosaicAttnParams(nn.Module):
    def __init__(self, d, eps, qk_ln):
        super().__init__()
        self.Wqkv = nn.Linear(d, 3 * d, bias=False)
        if qk_ln:
            self.q_ln = _LNNoBias(d, eps)
            self.k_ln = _LNNoBias(d, eps)
        self.out_proj = nn.Linear(d, d, bias=False)


class _MosaicMLPParams(nn.Module):
    def __init__(self, d, r):
        super().__init__()
        self.mlp_up = nn.Linear(d, r * d, bias=False)
        self.mlp_down = nn.Linear(r * d, d, bias=False)


class MosaicGPTBlock(nn.Module):
    """mosaic_gpt GPTBlock: x += out_proj(attn(k_ln / q_ln(Wqkv(ln_1(x)))));  x += mlp_down(gelu(mlp_up(ln_2(x))))"""

    def __init__(self, c):
        super().__init__()
        self.c = c
        self.ln_1 = _LNNoBias(c.d_model, c.layer_norm_epsilon)
        self.attn = _MosaicAttnParams(c.d_model, c.layer_norm_epsilon, c.attn_qk_ln)
        self.ln_2 = _LNNoBias(c.d_model, c.layer_norm_epsilon)
        self.mlp = _MosaicMLPParams(c.d_model, c.mlp_ratio)

    def forward(self, x, attention_mask=None, alibi=None, cache=None, pos0=0, **kw):
        c, a, f = self.c, self.attn, self.mlp
        qk = (a.q_ln.weight, a.k_ln.weight, a.q_ln.eps) if c.attn_qk_ln else None
        if cache is not None:
            x = F_.self_attn_block_cached(x, self.ln_1.weight, None, a.Wqkv.weight, None, a.out_proj.weight, None, c.n_heads,
                                          cache, pos0, interleaved=False, eps=self.ln_1.eps, alibi=alibi, qk_ln=qk)
        else:
            x = F_.self_attn_block(x, self.ln_1.weight, None, a.Wqkv.weight, None, a.out_proj.weight, None, c.n_heads,
                                   kv_len=attention_mask, interleaved=False, causal=True, eps=self.ln_1.eps, alibi=alibi, qk_ln=qk)
        return F_.mlp_block(x, self.ln_2.weight, None, f.mlp_up.weight, None, f.mlp_down.weight, None, "gelu", eps=self.ln_2.eps)


class _MosaicBody(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.wte = nn.Embedding(c.vocab_size, c.d_model)
        self.blocks = nn.ModuleList([MosaicGPTBlock(c) for _ in range(c.n_layers)])
        self.ln_f = _LNNoBias(c.d_model, c.layer_norm_epsilon)

    @property
    def norm_f(self):          # MptForCausalLM.forward's name for the final norm
        return self.ln_f


class MosaicGPT(MptForCausalLM):
    """the MPT-1B tower: MptForCausalLM's forward over mosaic_gpt's module names (``ln_1 / attn.{Wqkv,q_ln,k_ln,out_proj} / ln_2 /
    mlp.{mlp_up,mlp_down} / ln_f``; logits = h wte^T, no separate head parameter in the checkpoint)."""

    def __init__(self, config):
        _TowerBase.__init__(self)
        self.config = config
        self.supports_packed = not config.attn_qk_ln          # SelfAttnBlockFn's packed form has no QK-LayerNorm
        self.transformer = _MosaicBody(config)
        self.lm_head = nn.Linear(config.d_model, config.vocab_size, bias=False)
        self.lm_head.weight = self.transformer.wte.weight
        self._slopes = None


def build_lm(name_or_config):
    if isinstance(name_or_config, _Cfg):
        c = name_or_config
    else:
        key = [k for k in LM_CONFIGS if k.lower() in str(name_or_config).lower()]
        if not key:
            raise ValueError(f"unknown lang_encoder_path {name_or_config!r}; known: {sorted(LM_CONFIGS)} (or pass a config object)")
        c = LM_CONFIGS[key[0]]()
    return {"gpt_neox": GPTNeoXForCausalLM, "opt": OPTForCausalLM, "llama": LlamaForCausalLM, "mpt": MptForCausalLM,
            "mosaic_gpt": MosaicGPT}[c.model_type](c)
