"""Shared helpers for the end-to-end parity tests / smoke(): build the CPU oracle and the HIP model with the
same (bf16-representable) weights and the same synthetic batch.  TEST INFRASTRUCTURE (imports oracle/)."""
import torch

from oracle import flamingo as ofl, lm as olm, mpt as ompt, vit as ovit, train_step as ots

bf16 = torch.bfloat16

TINY = dict(vit=dict(image_size=32, patch_size=8, width=128, layers=2, heads=2, mlp_dim=256, output_dim=64),
            lm=dict(kind="neox", vocab_size=512, hidden_size=160, num_hidden_layers=4, num_attention_heads=2,
                    intermediate_size=320), every=2, T=3, L=48, B=2, n_items=40, base_vocab=300)
TINY_OPT = dict(vit=dict(image_size=32, patch_size=16, width=64, layers=1, heads=1, mlp_dim=128, output_dim=32),
                lm=dict(kind="opt", vocab_size=512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, ffn_dim=256,
                        max_position_embeddings=128), every=1, T=2, L=40, B=2, n_items=40, base_vocab=300)
TINY_PAR = dict(vit=dict(image_size=32, patch_size=8, width=128, layers=1, heads=2, mlp_dim=256, output_dim=64),
                lm=dict(kind="neox", vocab_size=512, hidden_size=256, num_hidden_layers=2, num_attention_heads=2,
                        intermediate_size=512, rotary_pct=0.25, use_parallel_residual=True), every=1, T=2, L=40, B=2,
                n_items=40, base_vocab=300)
TINY_MPT = dict(vit=dict(image_size=32, patch_size=8, width=128, layers=1, heads=2, mlp_dim=256, output_dim=64),
                lm=dict(kind="mpt", vocab_size=512, d_model=192, n_layers=2, n_heads=3), every=1, T=2, L=40, B=2,
                n_items=40, base_vocab=300)          # 3 heads of 64: non-power-of-two ALiBi slopes, tied head
# the "3b" towers (mmrec.py:475-494): MPT-1B = mosaic_gpt with LayerNorm over q and over k (4 heads of 64, tied head)
TINY_MOSAIC = dict(vit=dict(image_size=32, patch_size=8, width=128, layers=1, heads=2, mlp_dim=256, output_dim=64),
                   lm=dict(kind="mosaic", vocab_size=512, d_model=256, n_layers=2, n_heads=4), every=1, T=2, L=40, B=2,
                   n_items=40, base_vocab=300)
# every frozen Linear has both dimensions % 128 == 0 (the MX-fp8 path's tile contract): 2 heads of 128, ALiBi, tied head
TINY_MX = dict(vit=dict(image_size=32, patch_size=8, width=128, layers=2, heads=2, mlp_dim=256, output_dim=64),
               lm=dict(kind="mpt", vocab_size=512, d_model=256, n_layers=2, n_heads=2), every=1, T=2, L=48, B=2,
               n_items=40, base_vocab=300)
# BASELINE cfg2 / cfg3 at FULL WIDTH and reduced depth -- the model bench.cpu_baseline's bounded sample uses: ViT-L/14 widths
# with 3 of 24 blocks, GPT-NeoX-3B widths (H 2560, 32 heads of 80, FFN 10 240) with 4 of 32 layers and 2 of 16 gated
# cross-attention blocks, Perceiver 1 of 6 layers, V = 74 053, T = 8, L = 512.  The fp32 oracle runs it in tens of seconds.
CFG2_SLIM = dict(vit=dict(image_size=224, patch_size=14, width=1024, layers=3, heads=16, mlp_dim=4096, output_dim=768),
                 lm=dict(kind="neox", vocab_size=0, hidden_size=2560, num_hidden_layers=4, num_attention_heads=32,
                         intermediate_size=10240), every=2, T=8, L=512, B=2, n_items=22738, base_vocab=50277,
                 n_img_tokens=1024, perceiver_depth=1, std=0.02, min_fill=0.75)
# BASELINE cfg4 (the H&M path, unimp_hm.sh shapes) at full width, reduced depth: 16 history images per user (1024 media latents
# = the longest segment-masked key range), 14 901 items -> V = 66 216, gamma-2 focal loss
CFG4_SLIM = dict(vit=dict(image_size=224, patch_size=14, width=1024, layers=2, heads=16, mlp_dim=4096, output_dim=768),
                 lm=dict(kind="neox", vocab_size=0, hidden_size=2560, num_hidden_layers=4, num_attention_heads=32,
                         intermediate_size=10240), every=2, T=16, L=512, B=2, n_items=14901, base_vocab=50277,
                 n_img_tokens=1024, perceiver_depth=1, std=0.02, min_fill=0.8)
# BASELINE cfg5's model family at full width, reduced depth, on an image-token-generation shaped batch: MPT-7B widths
# (d_model 4096, 32 heads of 128, ALiBi, tied head) with 2 of 32 blocks and one gated cross-attention block, L = 1024, T = 2
CFG5_SLIM = dict(vit=dict(image_size=224, patch_size=14, width=1024, layers=2, heads=16, mlp_dim=4096, output_dim=768),
                 lm=dict(kind="mpt", vocab_size=0, d_model=4096, n_layers=2, n_heads=32), every=2, T=2, L=1024, B=1,
                 n_items=22738, base_vocab=50277, n_img_tokens=1024, perceiver_depth=1, std=0.02, min_fill=0.9)


def build_oracle(cfg, gate=0.5, seed=0):
    from unimp_amd.synthetic import TokenLayout
    torch.manual_seed(seed)
    layout = TokenLayout(cfg["base_vocab"], cfg["n_items"], cfg.get("n_img_tokens", 16))
    lmc = dict(cfg["lm"])
    kind = lmc.pop("kind")
    lmc["vocab_size"] = layout.vocab
    v = ovit.VisionTransformer(**cfg["vit"])
    if kind == "neox":
        lm = olm.GPTNeoXForCausalLM(olm.NeoXConfig(**lmc))
    elif kind == "mpt":
        lm = ompt.MptForCausalLM(ompt.MPTConfig(**lmc))
    elif kind == "mosaic":
        lm = ompt.MosaicGPT(ompt.MosaicGPTConfig(**lmc))
    else:
        lm = olm.OPTForCausalLM(olm.OPTConfig(**lmc))
    m = ofl.Flamingo(v, lm, layout.eoc, layout.media, vis_dim=cfg["vit"]["width"], cross_attn_every_n_layers=cfg["every"])
    if "perceiver_depth" in cfg:
        m.perceiver = ofl.PerceiverResampler(dim=cfg["vit"]["width"], depth=cfg["perceiver_depth"])
    for n, p in m.named_parameters():
        if p.dim() >= 2 and "embedding" not in n and "latents" not in n:
            p.data.normal_(0, cfg.get("std", 0.06))
        elif "bias" in n:
            p.data.normal_(0, 0.02)
    for g in m.lang_encoder.gated_cross_attn_layers:
        if g is not None:
            g.attn_gate.data.fill_(gate)
            g.ff_gate.data.fill_(-gate)
    for p in m.parameters():                       # make every weight bf16-representable
        p.data = p.data.to(bf16).float()
    ofl.freeze_like_factory(m)
    lm.get_output_embeddings().weight.requires_grad_(True)     # head re-created by resize_token_embeddings
    return m, layout


def build_hip(cfg, oracle_model, layout, device="cuda"):
    from unimp_amd.flamingo import Flamingo, PerceiverResampler, freeze_like_factory
    from unimp_amd.lm import build_lm, NeoXConfig, OPTConfig, MPTConfig, MosaicGPTConfig
    from unimp_amd.vit import VisionTransformer, CLIPStub
    lmc = dict(cfg["lm"])
    kind = lmc.pop("kind")
    lmc["vocab_size"] = layout.vocab
    with torch.device(device):
        v = VisionTransformer(**cfg["vit"])
        lm = build_lm({"neox": NeoXConfig, "opt": OPTConfig, "mpt": MPTConfig, "mosaic": MosaicGPTConfig}[kind](**lmc))
        m = Flamingo(CLIPStub(v), lm, layout.eoc, layout.media, vis_dim=cfg["vit"]["width"], cross_attn_every_n_layers=cfg["every"])
        if "perceiver_depth" in cfg:
            m.perceiver = PerceiverResampler(dim=cfg["vit"]["width"], depth=cfg["perceiver_depth"])
    m.to(dtype=bf16)
    missing, unexpected = m.load_state_dict(oracle_model.state_dict(), strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    freeze_like_factory(m)
    m.lang_encoder.get_output_embeddings().weight.requires_grad_(True)
    return m


def make_batch(cfg, layout, seed=1234):
    from unimp_amd.synthetic import make_batch as mb
    b = mb(layout, cfg["B"], cfg["T"], cfg["L"], image_size=cfg["vit"]["image_size"], seed=seed, min_fill=cfg.get("min_fill", 0.6))
    b["vision_x"] = b["vision_x"].to(bf16).float()          # bf16-representable pixels
    return b


def oracle_step(m, layout, batch, gamma=2.0, use_reweight=True):
    """forward + weighted focal loss + backward on the CPU oracle; returns (logits, loss, labels, grads dict)."""
    sp = layout.special()
    labels = torch.from_numpy(ots.label_mask_loop(batch["lang_x"].numpy(), sp["answer_id"], sp["eoc_id"], sp["pad_id"], sp["media_id"]))
    m.zero_grad()
    out = m(batch["vision_x"], batch["lang_x"], batch["attention_mask"], labels=None)
    loss = ots.weighted_focal_ce(out["logits"], labels, batch["weights"], gamma, use_reweight)
    loss.backward()
    grads = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    return out["logits"].detach(), loss.detach(), labels, grads


def rel_l2(got, want):
    got, want = got.float().cpu(), want.float().cpu()
    return ((got - want).norm() / (want.norm() + 1e-12)).item()


def bf16_noise_floor(m, layout, batch, labels, ref_grads, gamma=2.0, use_reweight=True):
    """per-parameter rel-L2 deviation of the oracle's own gradients when the oracle is re-run under bf16 autocast."""
    m.zero_grad()
    with torch.autocast("cpu", dtype=bf16):
        out = m(batch["vision_x"], batch["lang_x"], batch["attention_mask"])
        loss = ots.weighted_focal_ce(out["logits"].float(), labels, batch["weights"], gamma, use_reweight)
    loss.backward()
    return {n: (rel_l2(p.grad, ref_grads[n]) if ref_grads[n].norm() > 0 else 0.0)
            for n, p in m.named_parameters() if p.grad is not None and n in ref_grads}


def bf16_logit_floor(m, batch):
    """rel-L2 deviation of the oracle's own logits when the SAME fp32 oracle is re-run under bf16 autocast."""
    with torch.no_grad():
        want = m(batch["vision_x"], batch["lang_x"], batch["attention_mask"])["logits"]
        with torch.autocast("cpu", dtype=bf16):
            low = m(batch["vision_x"], batch["lang_x"], batch["attention_mask"])["logits"]
    return rel_l2(low, want)


def argmax_agreement(got, want, valid):
    """north_star's "token-id argmax bit-exact", as a number.  got / want [B, L, V] logits, valid [B, L] bool (real, unpadded
    positions).  Returns dict(rate = share of ALL valid positions with the same argmax, sure_rate = share of positions where
    the HIP path's own top-2 margin exceeds 8 sigma of its measured logit error (sigma = RMS(got - want) over valid rows),
    sure_equal = argmax identical on every such position)."""
    got, want = got.float().cpu(), want.float().cpu()
    g, w = got[valid], want[valid]
    sigma = (g - w).pow(2).mean().sqrt().item()
    top2 = g.topk(2, -1).values
    sure = (top2[:, 0] - top2[:, 1]) > 8 * sigma
    ga, wa = g.argmax(-1), w.argmax(-1)
    return dict(rate=(ga == wa).float().mean().item(), n=int(valid.sum()), sigma=sigma, sure_rate=sure.float().mean().item(),
                sure_equal=bool(torch.equal(ga[sure], wa[sure])), n_sure=int(sure.sum()))
