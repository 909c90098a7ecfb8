"""Architecture known-answer tests for the (unpinned) Flamingo restatement."""
import torch
import pytest
from oracle import flamingo as ofl, lm as olm, vit as ovit

IMG, EOC = 120, 121


def build(gate=0.5, every=1, seed=0):
    torch.manual_seed(seed)
    v = ovit.VisionTransformer(image_size=16, patch_size=8, width=32, layers=1, heads=2, mlp_dim=64, output_dim=8)
    c = olm.NeoXConfig(vocab_size=128, hidden_size=48, num_hidden_layers=2, num_attention_heads=4,
                       intermediate_size=96)
    m = ofl.Flamingo(v, olm.GPTNeoXForCausalLM(c), EOC, IMG, vis_dim=32, cross_attn_every_n_layers=every)
    for g in m.lang_encoder.gated_cross_attn_layers:
        if g is not None:
            g.attn_gate.data.fill_(gate)
            g.ff_gate.data.fill_(gate)
    return m.eval()


def batch(T=3, L=20, seed=1):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, 100, (2, L), generator=g)
    ids[:, [3, 8, 14]] = IMG
    vx = torch.randn(2, T, 1, 3, 16, 16, generator=g)
    return vx, ids


def test_gate_zero_is_plain_lm():
    m = build(gate=0.0)
    vx, ids = batch()
    with torch.no_grad():
        a = m(vx, ids)["logits"]
        b = m.lang_encoder.gpt_neox.embed_in(ids)
        bias = olm._mask_bias(None, ids.shape[1], b.dtype)
        for l in m.lang_encoder.old_decoder_blocks:
            b = l(b, attention_mask=bias)
        b = m.lang_encoder.embed_out(m.lang_encoder.gpt_neox.final_layer_norm(b))
    assert torch.equal(a, b)
    m2 = build(gate=0.5)
    with torch.no_grad():
        assert (m2(vx, ids)["logits"] - b).abs().max() > 1e-3


def test_text_before_first_image_untouched_and_image_locality():
    m = build(gate=0.5)
    vx, ids = batch()
    with torch.no_grad():
        a = m(vx, ids)["logits"]
        vx2 = vx.clone()
        vx2[:, 1] += 1.0                       # perturb image #2 (attended from position 8 on)
        b = m(vx2, ids)["logits"]
        vx3 = vx.clone()
        vx3[:, 0] += 1.0
        c = m(vx3, ids)["logits"]
    assert torch.equal(a[:, :8], b[:, :8]) and not torch.equal(a[:, 8:], b[:, 8:])
    assert torch.equal(a[:, :3], c[:, :3]) and not torch.equal(a[:, 3:], c[:, 3:])


def test_perceiver_permutation_equivariance():
    m = build()
    x = torch.randn(2, 3, 1, 4, 32)
    with torch.no_grad():
        y = m.perceiver(x)
        y2 = m.perceiver(x.flip(1))
    assert torch.allclose(y.flip(1), y2, atol=1e-6)


def test_param_names_and_freeze():
    m = build(every=2)
    ofl.freeze_like_factory(m)
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    assert any("gated_cross_attn_layer" in n for n in names)
    assert "perceiver.latents" in names and "lang_encoder.gpt_neox.embed_in.weight" in names
    assert not any(n.startswith("vision_encoder") for n in names)
    sd = m.state_dict()
    assert "lang_encoder.gated_cross_attn_layers.1.attn.to_kv.weight" in sd
    assert "lang_encoder.gpt_neox.layers.1.gated_cross_attn_layer.ff.1.weight" in sd
    assert "lang_encoder.old_decoder_blocks.0.attention.dense.weight" in sd
    assert m.lang_encoder.gated_cross_attn_layers[0] is None
