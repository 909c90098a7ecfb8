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


def build_oracle(cfg, gate=0.5, seed=0):
    from unimp_amd.synthetic import TokenLayout
    torch.manual_seed(seed)
    layout = TokenLayout(cfg["base_vocab"], cfg["n_items"], 16)
    lmc = dict(cfg["lm"])
    kind = lmc.pop("kind")
    lmc["vocab_size"] = layout.vocab
    v = ovit.VisionTransformer(**cfg["vit"])
    if kind == "neox":
        lm = olm.GPTNeoXForCausalLM(olm.NeoXConfig(**lmc))
    elif kind == "mpt":
        lm = ompt.MptForCausalLM(ompt.MPTConfig(**lmc))
    else:
        lm = olm.OPTForCausalLM(olm.OPTConfig(**lmc))
    m = ofl.Flamingo(v, lm, layout.eoc, layout.media, vis_dim=cfg["vit"]["width"], cross_attn_every_n_layers=cfg["every"])
    for n, p in m.named_parameters():
        if p.dim() >= 2 and "embedding" not in n and "latents" not in n:
            p.data.normal_(0, 0.06)
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
    from unimp_amd.flamingo import Flamingo, freeze_like_factory
    from unimp_amd.lm import build_lm, NeoXConfig, OPTConfig, MPTConfig
    from unimp_amd.vit import VisionTransformer, CLIPStub
    lmc = dict(cfg["lm"])
    kind = lmc.pop("kind")
    lmc["vocab_size"] = layout.vocab
    with torch.device(device):
        v = VisionTransformer(**cfg["vit"])
        lm = build_lm({"neox": NeoXConfig, "opt": OPTConfig, "mpt": MPTConfig}[kind](**lmc))
        m = Flamingo(CLIPStub(v), lm, layout.eoc, layout.media, vis_dim=cfg["vit"]["width"], cross_attn_every_n_layers=cfg["every"])
    m.to(dtype=bf16)
    missing, unexpected = m.load_state_dict(oracle_model.state_dict(), strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    freeze_like_factory(m)
    m.lang_encoder.get_output_embeddings().weight.requires_grad_(True)
    return m


def make_batch(cfg, layout, seed=1234):
    from unimp_amd.synthetic import make_batch as mb
    b = mb(layout, cfg["B"], cfg["T"], cfg["L"], image_size=cfg["vit"]["image_size"], seed=seed, min_fill=0.6)
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
