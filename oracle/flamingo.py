"""Oracle: Flamingo / Perceiver / gated cross-attention (fp32, CPU).  TEST INFRASTRUCTURE ONLY.

The reference imports these from the pip package ``open-flamingo==2.0.1``
(requirements.txt:36; ``from open_flamingo import create_model_and_transforms``, ``Flamingo``
UniMP/mmrec.py:20-22; construction UniMP/mmrec.py:476-524; call UniMP/mmrec.py:177-181).
Its source is NOT under /root/reference and the package is not installed here, so this is
a restatement of the published algorithm (open_flamingo/src/{flamingo,flamingo_lm,helpers}.py,
as recorded in SURVEY.md Appendix A.1-A.6).  PARITY UNPINNED at this boundary; anchored by
the architecture known-answer tests in tests/test_oracle_kat.py.
Module tree and parameter names match upstream so OpenFlamingo ``checkpoint.pt`` /
UniMP ``weights_epoch_*.pt`` key sets line up (A.6).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .numerics import st


class _FF(nn.Sequential):
    """LayerNorm -> Linear -> GELU -> Linear with the storage points of oracle/numerics.py (identity by default)."""

    def forward(self, x):
        return self[3](st("act", self[2](self[1](st("ln", self[0](x))))))


def FeedForward(dim, mult=4):
    # indices 0,1,2,3 matter for parameter names (A.2)
    return _FF(nn.LayerNorm(dim), nn.Linear(dim, dim * mult, bias=False), nn.GELU(),
               nn.Linear(dim * mult, dim, bias=False))


class PerceiverAttention(nn.Module):
    def __init__(self, dim, dim_head=64, heads=8):
        super().__init__()
        self.scale, self.heads = dim_head ** -0.5, heads
        inner = dim_head * heads
        self.norm_media, self.norm_latents = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)

    def forward(self, x, latents):
        # x (b,T,n1,D)  latents (b,T,n2,D)
        x, latents = st("ln", self.norm_media(x)), st("ln", self.norm_latents(latents))
        h = self.heads
        q = st("gemm", self.to_q(latents))
        kv_in = torch.cat((x, latents), -2)
        k, v = st("gemm", self.to_kv(kv_in)).chunk(2, -1)
        sp = lambda t: t.view(*t.shape[:3], h, -1).permute(0, 3, 1, 2, 4)   # b h T n d
        q, k, v = sp(q) * self.scale, sp(k), sp(v)
        sim = q @ k.transpose(-1, -2)
        sim = sim - sim.amax(-1, keepdim=True).detach()
        out = st("attn_p", sim.softmax(-1)) @ v
        out = st("attn_o", out.permute(0, 2, 3, 1, 4).flatten(-2))           # b T n (h d)
        return self.to_out(out)


class PerceiverResampler(nn.Module):
    def __init__(self, dim, depth=6, dim_head=64, heads=8, num_latents=64, ff_mult=4):
        super().__init__()
        self.latents = nn.Parameter(torch.randn(num_latents, dim))
        self.layers = nn.ModuleList([nn.ModuleList([PerceiverAttention(dim, dim_head, heads),
                                                    FeedForward(dim, ff_mult)]) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim)

    def forward(self, x):
        # x (b,T,F,v,D) -> (b,T,n,D)
        b, T, Fr, v, D = x.shape
        x = x.reshape(b, T, Fr * v, D)
        lat = self.latents[None, None].expand(b, T, -1, -1)
        for attn, ff in self.layers:
            lat = st("res", attn(x, lat) + lat)
            lat = st("res", ff(lat) + lat)
        return st("vis", self.norm(lat))


class MaskedCrossAttention(nn.Module):
    def __init__(self, dim, dim_visual, dim_head=64, heads=8, only_attend_immediate_media=True):
        super().__init__()
        self.scale, self.heads = dim_head ** -0.5, heads
        inner = dim_head * heads
        self.norm = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim_visual, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)
        self.only_attend_immediate_media = only_attend_immediate_media

    def forward(self, x, media, media_locations=None, use_cached_media=False):
        # x (B,L,D)  media (B,T,n,Dv)  media_locations (B,L) bool
        B, T, n = media.shape[:3]
        h = self.heads
        x = st("ln", self.norm(x))
        q = st("gemm", self.to_q(x))
        media = media.reshape(B, T * n, -1)
        k, v = st("gemm", self.to_kv(media)).chunk(2, -1)
        sp = lambda t: t.view(B, t.shape[1], h, -1).transpose(1, 2)
        q, k, v = sp(q) * self.scale, sp(k), sp(v)
        sim = q @ k.transpose(-1, -2)                                        # B h L (T n)
        if media_locations is not None:
            media_time = torch.arange(T) + 1
            if use_cached_media:
                text_time = media_locations.sum(-1, keepdim=True).expand(-1, x.shape[1])
            else:
                text_time = media_locations.cumsum(-1)
            op = torch.eq if self.only_attend_immediate_media else torch.ge
            mask = op(text_time[:, None, :, None], media_time.repeat_interleave(n)[None, None, None, :])
            sim = sim.masked_fill(~mask, -torch.finfo(sim.dtype).max)
        sim = sim - sim.amax(-1, keepdim=True).detach()
        attn = st("attn_p", sim.softmax(-1))
        if media_locations is not None and self.only_attend_immediate_media:
            attn = attn.masked_fill((text_time == 0)[:, None, :, None], 0.0)   # rows with no image yet
        out = st("attn_o", (attn @ v).transpose(1, 2).reshape(B, -1, h * v.shape[-1]))
        return self.to_out(out)


class GatedCrossAttentionBlock(nn.Module):
    def __init__(self, dim, dim_visual, dim_head=64, heads=8, ff_mult=4, only_attend_immediate_media=True):
        super().__init__()
        self.attn = MaskedCrossAttention(dim, dim_visual, dim_head, heads, only_attend_immediate_media)
        self.attn_gate = nn.Parameter(torch.tensor([0.0]))
        self.ff = FeedForward(dim, ff_mult)
        self.ff_gate = nn.Parameter(torch.tensor([0.0]))

    def forward(self, x, media, media_locations=None, use_cached_media=False):
        x = st("res", self.attn(x, media, media_locations, use_cached_media) * self.attn_gate.tanh() + x)
        return st("res", self.ff(x) * self.ff_gate.tanh() + x)


class FlamingoLayer(nn.Module):
    def __init__(self, gated_cross_attn_layer, decoder_layer):
        super().__init__()
        self.gated_cross_attn_layer = gated_cross_attn_layer
        self.decoder_layer = decoder_layer
        self.vis_x = self.media_locations = None
        self.use_cached_media = False

    def is_conditioned(self):
        return self.vis_x is not None and self.media_locations is not None

    def condition_vis_x(self, v): self.vis_x = v
    def condition_media_locations(self, m): self.media_locations = m
    def condition_use_cached_media(self, u): self.use_cached_media = u

    def forward(self, lang_x, attention_mask=None, **kw):
        if self.gated_cross_attn_layer is not None:
            assert self.vis_x is not None and self.media_locations is not None
            lang_x = self.gated_cross_attn_layer(lang_x, self.vis_x, self.media_locations,
                                                 self.use_cached_media)
        return self.decoder_layer(lang_x, attention_mask=attention_mask, **kw)


def _getattr_path(o, path):
    for p in path.split("."):
        o = getattr(o, p)
    return o


def _setattr_path(o, path, v):
    parts = path.split(".")
    for p in parts[:-1]:
        o = getattr(o, p)
    setattr(o, parts[-1], v)


def init_flamingo(lang_encoder, media_token_id, lang_hidden_size, vis_hidden_size, cross_attn_every_n_layers):
    """FlamingoLMMixin.init_flamingo (A.5): splice GatedCrossAttentionBlocks into the tower."""
    attr = lang_encoder.decoder_layers_attr
    old = _getattr_path(lang_encoder, attr)
    lang_encoder.old_decoder_blocks = old
    lang_encoder.gated_cross_attn_layers = nn.ModuleList([
        GatedCrossAttentionBlock(lang_hidden_size, vis_hidden_size)
        if (i + 1) % cross_attn_every_n_layers == 0 else None for i in range(len(old))])
    _setattr_path(lang_encoder, attr, nn.ModuleList(
        [FlamingoLayer(g, d) for g, d in zip(lang_encoder.gated_cross_attn_layers, old)]))
    lang_encoder.media_token_id = media_token_id
    lang_encoder._use_cached_vision_x = False


class Flamingo(nn.Module):
    def __init__(self, vision_encoder, lang_encoder, eoc_token_id, media_token_id, vis_dim,
                 cross_attn_every_n_layers=1):
        super().__init__()
        self.eoc_token_id, self.media_token_id, self.vis_dim = eoc_token_id, media_token_id, vis_dim
        self.lang_dim = lang_encoder.config.hidden_size
        self.vision_encoder = vision_encoder
        self.perceiver = PerceiverResampler(dim=vis_dim)
        self.lang_encoder = lang_encoder
        init_flamingo(lang_encoder, media_token_id, self.lang_dim, vis_dim, cross_attn_every_n_layers)

    def _layers(self):
        return _getattr_path(self.lang_encoder, self.lang_encoder.decoder_layers_attr)

    def forward(self, vision_x, lang_x, attention_mask=None, labels=None,
                clear_conditioned_layers=True, past_key_values=None, use_cache=False):
        assert vision_x.ndim == 6, "vision_x should be of shape (b, T_img, F, C, H, W)"
        b, T, Fr = vision_x.shape[:3]
        assert Fr == 1, "Only single frame supported"
        with torch.no_grad():
            tok = st("vis", self.vision_encoder(vision_x.flatten(0, 2))[1])
        vis = self.perceiver(tok.view(b, T, Fr, *tok.shape[1:]))
        media_locations = lang_x == self.media_token_id
        for layer in self._layers():
            layer.condition_vis_x(vis)
            layer.condition_media_locations(media_locations)
            layer.condition_use_cached_media(False)
        out = self.lang_encoder(input_ids=lang_x, attention_mask=attention_mask, labels=labels)
        if clear_conditioned_layers:
            for layer in self._layers():
                layer.condition_vis_x(None)
                layer.condition_media_locations(None)
        return out


def freeze_like_factory(model):
    """create_model_and_transforms (A.5): freeze all, unfreeze perceiver, gated xattn, input embeddings."""
    model.requires_grad_(False)
    model.perceiver.requires_grad_(True)
    model.lang_encoder.gated_cross_attn_layers.requires_grad_(True)
    model.lang_encoder.get_input_embeddings().requires_grad_(True)
