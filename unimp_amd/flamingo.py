"""Flamingo / PerceiverResampler / GatedCrossAttentionBlock / FlamingoLayer on the HIP kernels.

Drop-in for the ``open_flamingo`` classes UniMP imports (UniMP/mmrec.py:20-22): same module tree, parameter
names, constructor arguments and ``forward`` / attribute surface that mmrec.py touches
(``.vision_encoder``, ``.perceiver``, ``.lang_encoder.gated_cross_attn_layers``, ``.eoc_token_id``,
``.media_token_id``, ``forward(vision_x, lang_x, attention_mask, labels, ...)`` returning an object indexable by
``[0]`` and ``["logits"]``: mmrec.py:177-190).  Semantics restated from open-flamingo 2.0.1.
"""
import torch
import torch.nn as nn

from . import functional as F_
from . import ops

bf16 = torch.bfloat16


def FeedForward(dim, mult=4):
    """Parameter container with upstream indices (0 = LayerNorm, 1 = Linear, 3 = Linear); run via functional.mlp_block."""
    return nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, dim * mult, bias=False), nn.GELU(),
                         nn.Linear(dim * mult, dim, bias=False))


def _ff_block(ff, x, gate=None):
    return F_.mlp_block(x, ff[0].weight, ff[0].bias, ff[1].weight, None, ff[3].weight, None, "gelu", gate=gate, eps=ff[0].eps)


class PerceiverAttention(nn.Module):
    def __init__(self, *, dim, dim_head=64, heads=8):
        super().__init__()
        self.scale, self.heads = dim_head ** -0.5, heads
        inner = dim_head * heads
        self.norm_media, self.norm_latents = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)

    def forward(self, x, latents):
        """x [G, n1, D], latents [G, n2, D] -> latents + attention (residual fused)."""
        nm, nl = self.norm_media, self.norm_latents
        return F_.perceiver_attn(x, latents, nm.weight, nm.bias, nl.weight, nl.bias, self.to_q.weight, self.to_kv.weight,
                                 self.to_out.weight, self.heads, nm.eps)


class PerceiverResampler(nn.Module):
    def __init__(self, *, dim, depth=6, dim_head=64, heads=8, num_latents=64, max_num_media=None, max_num_frames=None,
                 ff_mult=4):
        super().__init__()
        if max_num_media is not None or max_num_frames is not None:
            raise NotImplementedError("media/frame position embeddings are not used by UniMP (defaults None)")
        self.latents = nn.Parameter(torch.randn(num_latents, dim))
        self.layers = nn.ModuleList([nn.ModuleList([PerceiverAttention(dim=dim, dim_head=dim_head, heads=heads),
                                                    FeedForward(dim, ff_mult)]) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim)

    def forward(self, x):
        """x (b, T, F, v, D) -> (b, T, n, D)"""
        b, T, Fr, v, D = x.shape
        G = b * T
        xm = x.reshape(G, Fr * v, D)
        n = self.latents.shape[0]
        lat = F_.BcastRowsFn.apply(self.latents, G).view(G, n, D)
        for attn, ff in self.layers:
            lat = attn(xm, lat)
            lat = _ff_block(ff, lat)
        return F_.layer_norm(lat, self.norm.weight, self.norm.bias, self.norm.eps).view(b, T, n, D)


class MaskedCrossAttention(nn.Module):
    def __init__(self, *, dim, dim_visual, dim_head=64, heads=8, only_attend_immediate_media=True):
        super().__init__()
        if not only_attend_immediate_media:
            raise NotImplementedError("only_attend_immediate_media=False is not used by open_flamingo's FlamingoLayer")
        self.scale, self.heads = dim_head ** -0.5, heads
        inner = dim_head * heads
        self.norm = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim_visual, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)


class GatedCrossAttentionBlock(nn.Module):
    def __init__(self, *, dim, dim_visual, dim_head=64, heads=8, ff_mult=4, only_attend_immediate_media=True):
        super().__init__()
        self.attn = MaskedCrossAttention(dim=dim, dim_visual=dim_visual, dim_head=dim_head, heads=heads,
                                         only_attend_immediate_media=only_attend_immediate_media)
        self.attn_gate = nn.Parameter(torch.tensor([0.0]))
        self.ff = FeedForward(dim, ff_mult)
        self.ff_gate = nn.Parameter(torch.tensor([0.0]))

    def forward(self, x, media, media_locations=None, use_cached_media=False, media_time=None, cache=None):
        """x [B, L, D]; media [B, T, n, Dv]; media_time int32 [B, L] = cumsum(media_locations) (text_time)."""
        B, T, n = media.shape[:3]
        if media_time is None:
            if media_locations is None:
                raise ValueError("media_locations (or media_time) is required")
            media_time = (media_locations.sum(-1, keepdim=True).expand(-1, x.shape[1]) if use_cached_media
                          else media_locations.cumsum(-1)).to(torch.int32).contiguous()
        a = self.attn
        if cache is not None:
            x = F_.gated_xattn_cached(x, media.reshape(B, T * n, -1), media_time, a.norm.weight, a.norm.bias, a.to_q.weight,
                                      a.to_kv.weight, a.to_out.weight, self.attn_gate, a.heads, n, cache, a.norm.eps)
        else:
            x = F_.gated_xattn(x, media.reshape(B, T * n, -1), media_time, a.norm.weight, a.norm.bias, a.to_q.weight,
                               a.to_kv.weight, a.to_out.weight, self.attn_gate, a.heads, n, a.norm.eps)
        return _ff_block(self.ff, x, gate=self.ff_gate)


class FlamingoLayer(nn.Module):
    def __init__(self, gated_cross_attn_layer, decoder_layer, gradient_checkpointing=False):
        super().__init__()
        self.gated_cross_attn_layer = gated_cross_attn_layer
        self.decoder_layer = decoder_layer
        self.vis_x = self.media_locations = self.media_time = None
        self.use_cached_media = False

    def is_conditioned(self):
        return self.vis_x is not None and self.media_locations is not None

    def condition_vis_x(self, vis_x): self.vis_x = vis_x
    def condition_media_locations(self, m): self.media_locations = m
    def condition_media_time(self, t): self.media_time = t
    def condition_use_cached_media(self, u): self.use_cached_media = u

    def forward(self, lang_x, attention_mask=None, **decoder_layer_kwargs):
        if self.gated_cross_attn_layer is not None:
            if self.vis_x is None:
                raise ValueError("vis_x must be conditioned before forward pass")
            if self.media_locations is None:
                raise ValueError("media_locations must be conditioned before forward pass")
            lang_x = self.gated_cross_attn_layer(lang_x, self.vis_x, media_locations=self.media_locations,
                                                 use_cached_media=self.use_cached_media, media_time=self.media_time,
                                                 cache=decoder_layer_kwargs.get("cache"))
        return self.decoder_layer(lang_x, attention_mask=attention_mask, **decoder_layer_kwargs)


def _getattr_path(o, path):
    for p in path.split("."):
        o = getattr(o, p)
    return o


def _setattr_path(o, path, v):
    parts = path.split(".")
    for p in parts[:-1]:
        o = getattr(o, p)
    setattr(o, parts[-1], v)


def init_flamingo(lang_encoder, media_token_id, lang_hidden_size, vis_hidden_size, cross_attn_every_n_layers,
                  gradient_checkpointing=False):
    """FlamingoLMMixin.init_flamingo (A.5)."""
    attr = lang_encoder.decoder_layers_attr
    old = _getattr_path(lang_encoder, attr)
    lang_encoder.old_decoder_blocks = old
    lang_encoder.gated_cross_attn_layers = nn.ModuleList([
        GatedCrossAttentionBlock(dim=lang_hidden_size, dim_visual=vis_hidden_size)
        if (i + 1) % cross_attn_every_n_layers == 0 else None for i in range(len(old))])
    _setattr_path(lang_encoder, attr, nn.ModuleList(
        [FlamingoLayer(g, d, gradient_checkpointing) for g, d in zip(lang_encoder.gated_cross_attn_layers, old)]))
    lang_encoder.media_token_id = media_token_id
    lang_encoder.initialized_flamingo = True
    lang_encoder._use_cached_vision_x = False


class Flamingo(nn.Module):
    def __init__(self, vision_encoder, lang_encoder, eoc_token_id, media_token_id, vis_dim, cross_attn_every_n_layers=1,
                 gradient_checkpointing=False):
        super().__init__()
        self.eoc_token_id, self.media_token_id, self.vis_dim = eoc_token_id, media_token_id, vis_dim
        cfg = lang_encoder.config
        self.lang_dim = getattr(cfg, "d_model", None) or cfg.hidden_size
        self.vision_encoder = vision_encoder.visual if hasattr(vision_encoder, "visual") else vision_encoder
        self.perceiver = PerceiverResampler(dim=vis_dim)
        self.lang_encoder = lang_encoder
        init_flamingo(lang_encoder, media_token_id, self.lang_dim, vis_dim, cross_attn_every_n_layers, gradient_checkpointing)
        self._use_gradient_checkpointing = gradient_checkpointing

    def _layers(self):
        return _getattr_path(self.lang_encoder, self.lang_encoder.decoder_layers_attr)

    def forward(self, vision_x, lang_x, attention_mask=None, labels=None, clear_conditioned_layers=True,
                past_key_values=None, use_cache=False, head_rows=None):
        assert self.lang_encoder.initialized_flamingo, "Flamingo layers are not initialized. Please call `init_flamingo` first."
        assert self.lang_encoder._use_cached_vision_x or vision_x is not None, \
            "Must provide either vision_x or have precached media using cache_media()."
        if self.lang_encoder._use_cached_vision_x:
            assert vision_x is None and self.lang_encoder.is_conditioned()
            self._params_barrier()
        else:
            self._encode_vision_x(vision_x=vision_x)
            self._condition_media_locations(input_ids=lang_x)
        if past_key_values is not None:
            self._condition_cached_media(past_key_values, lang_x.shape[1])
        output = self.lang_encoder(input_ids=lang_x, attention_mask=attention_mask, labels=labels,
                                   past_key_values=past_key_values, use_cache=use_cache, head_rows=head_rows)
        if output.past_key_values is not None and output.past_key_values.media_count is None:
            output.past_key_values.media_count = (lang_x == self.media_token_id).sum(1, keepdim=True).to(torch.int32)
        if clear_conditioned_layers:
            self.clear_conditioned_layers()
        return output

    def _encode_vision_x(self, vision_x):
        assert vision_x.ndim == 6, "vision_x should be of shape (b, T_img, F, C, H, W)"
        b, T, Fr = vision_x.shape[:3]
        assert Fr == 1, "Only single frame supported"
        ve = self.vision_encoder
        prev = ve.output_tokens
        ve.output_tokens = True
        with torch.no_grad():
            tok = ve(vision_x.reshape(b * T * Fr, *vision_x.shape[3:]))[1]
        ve.output_tokens = prev
        self._params_barrier()            # everything above ran on frozen weights; from here on trainable parameters are read
        tok = tok.reshape(b, T, Fr, tok.shape[1], tok.shape[2])
        vis = self.perceiver(tok)
        for layer in self._layers():
            layer.condition_vis_x(vis)

    def _params_barrier(self):
        """train.Trainer(overlap_optimizer=True) runs clip + AdamW on a side stream and leaves its completion event here: the frozen
        ViT forward above overlaps the update, every reader of a trainable parameter comes after this wait (stream-ordered, no host block)"""
        ev = getattr(self, "_params_ready", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self._params_ready = None

    def _condition_media_locations(self, input_ids):
        media_locations = input_ids == self.media_token_id
        _, media_time = ops.label_mask(input_ids, -1, -1, -1, self.media_token_id, want_labels=False)
        for layer in self._layers():
            layer.condition_media_locations(media_locations)
            layer.condition_media_time(media_time)
            layer.condition_use_cached_media(False)

    def _repeat_conditioned_vision(self, k):
        """rows of the conditioned Perceiver output repeated k times (one copy per beam)."""
        layers = list(self._layers())
        vis = layers[0].vis_x.repeat_interleave(k, dim=0)
        for layer in layers:
            layer.condition_vis_x(vis)

    def _condition_cached_media(self, cache, n_new):
        """decode steps (FlamingoLMMixin.forward with ``use_cached_media_locations``, MaskedCrossAttention with
        ``use_cached_media``): every new token attends with text_time = #<image> in the prompt."""
        media_time = cache.media_count.expand(-1, n_new).contiguous()
        loc = media_time > 0                   # once per step, not once per layer (32 launches of a 4.5 us graph node each)
        for layer in self._layers():
            layer.condition_media_locations(loc)
            layer.condition_media_time(media_time)
            layer.condition_use_cached_media(True)

    def clear_conditioned_layers(self):
        for layer in self._layers():
            layer.condition_vis_x(None)
            layer.condition_media_locations(None)
            layer.condition_media_time(None)
            layer.condition_use_cached_media(None)

    @torch.no_grad()
    def generate(self, vision_x, lang_x, attention_mask=None, **kwargs):
        """open_flamingo ``Flamingo.generate`` (eval_rec.py:100-110): vision encoded once (repeated per beam),
        then greedy / beam search over the LM (generate.py: transformers' BeamSearchScorer semantics).  ``use_cache=True``
        (transformers' default) decodes with the KV cache: one prefill over the prompt, then one token per row and step,
        the cache rows following their beams, the step replayed as a HIP graph (decode.py; ``use_graph=False`` launches it
        kernel by kernel); ``use_cache=False`` re-scores the full sequences every step."""
        from .generate import beam_search, greedy_search
        from .decode import DecodeSession
        num_beams = kwargs.pop("num_beams", 1)
        # upstream repeats vision_x per beam BEFORE encoding it; the rows are identical, so the images are encoded once per
        # prompt here and the Perceiver output is repeated instead (same values row for row, 1/num_beams of the ViT work)
        eos_token_id = kwargs.pop("eos_token_id", self.eoc_token_id)
        pad_token_id = kwargs.pop("pad_token_id", eos_token_id)
        max_new_tokens = kwargs.pop("max_new_tokens", 20)
        nret = kwargs.pop("num_return_sequences", 1)
        early = kwargs.pop("early_stopping", False)
        ngram = kwargs.pop("no_repeat_ngram_size", 0)
        lp = kwargs.pop("length_penalty", 1.0)
        use_cache = kwargs.pop("use_cache", True)
        use_graph = kwargs.pop("use_graph", True)      # replay the decode step as one HIP graph (decode.py)
        trace = kwargs.pop("trace", None)              # tests: per-step candidate scores of the beam search
        if kwargs.pop("do_sample", False) or kwargs:
            raise NotImplementedError(f"unsupported generate() arguments: do_sample / {sorted(kwargs)}")
        was_training = self.training
        self.eval()
        lengths, seqs0 = None, lang_x
        if attention_mask is not None and not bool(attention_mask.all()):
            # a batch of prompts of different lengths, right-padded as collate_fn pads them (the reference evaluates one user
            # per call; several users per call share every weight read of the decode).  The KV-cache path runs every row at its
            # own position; the search bookkeeping sees the prompts left-padded, as transformers' generate would.
            if not use_cache:
                raise NotImplementedError("generate() with padded prompts needs use_cache=True")
            lengths = attention_mask.sum(1)
            L = lang_x.shape[1]
            if not bool((attention_mask == (torch.arange(L, device=lang_x.device)[None, :] < lengths[:, None])).all()):
                raise ValueError("generate(): attention_mask must be a right-padding mask")
            shift = (torch.arange(L, device=lang_x.device)[None, :] - (L - lengths)[:, None])
            seqs0 = torch.where(shift >= 0, lang_x.gather(1, shift.clamp(min=0)), torch.full_like(lang_x, pad_token_id))
        self.lang_encoder._use_cached_vision_x = True
        self._encode_vision_x(vision_x=vision_x)
        if num_beams > 1 and not use_cache:
            self._repeat_conditioned_vision(num_beams)
        try:
            session = [None]

            def logits_fn(seqs, src=None):
                if not use_cache:
                    self._condition_media_locations(input_ids=seqs)
                    return self.lang_encoder(input_ids=seqs, attention_mask=None, logits_last_only=True)["logits"][:, -1]
                if session[0] is None:
                    session[0] = DecodeSession(self, max_new_tokens, reorder=num_beams > 1, graph=use_graph, beams=num_beams)
                    return session[0].prefill(lang_x, lengths)
                return session[0].step(seqs[:, -1], src)
            if num_beams > 1:
                out = beam_search(logits_fn, seqs0, num_beams, max_new_tokens, eos_token_id, pad_token_id, nret, early, lp,
                                  ngram, stateful=True, trace=trace)
            else:
                out = greedy_search(logits_fn, seqs0, max_new_tokens, eos_token_id, pad_token_id)
        finally:
            self.clear_conditioned_layers()
            self.lang_encoder._use_cached_vision_x = False
            self.train(was_training)
        return out


def freeze_like_factory(model, freeze_lm_embeddings=False):
    """create_model_and_transforms: freeze all, unfreeze perceiver, gated xattn, (input embeddings)."""
    model.requires_grad_(False)
    model.perceiver.requires_grad_(True)
    model.lang_encoder.gated_cross_attn_layers.requires_grad_(True)
    if not freeze_lm_embeddings:
        model.lang_encoder.get_input_embeddings().requires_grad_(True)
