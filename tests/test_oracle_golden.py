"""Pin the CPU oracle against fixtures captured from the reference (oracle/make_golden.py)."""
import os
import numpy as np
import torch
import pytest

from oracle import train_step as ts
from oracle import vit as ovit
from oracle import lm as olm
from oracle import llama as ollama

G = os.path.join(os.path.dirname(__file__), "golden")


def _sd(z, prefix="sd."):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


@pytest.mark.parametrize("gamma", [0, 2])
@pytest.mark.parametrize("rw", [0, 1])
def test_train_step_vs_reference(gamma, rw):
    z = np.load(os.path.join(G, f"train_step_g{gamma}_rw{rw}.npz"))
    ANS, EOC, PAD, IMG = [int(v) for v in z["special"]]
    lab_loop = ts.label_mask_loop(z["ids"], ANS, EOC, PAD, IMG)
    lab_closed = ts.label_mask(z["ids"], ANS, EOC, PAD, IMG)
    assert np.array_equal(lab_loop, z["labels"])          # bit-exact integer work
    assert np.array_equal(lab_closed, z["labels"])
    logits = torch.from_numpy(z["logits"]).requires_grad_(True)
    loss = ts.weighted_focal_ce(logits, torch.from_numpy(z["labels"]), torch.from_numpy(z["weights"]),
                                gamma, bool(rw))
    assert abs(float(loss) - float(z["loss"])) <= 1e-6 * abs(float(z["loss"]))
    loss.backward()
    assert np.allclose(logits.grad.numpy(), z["dlogits"], rtol=1e-5, atol=1e-8)
    ana = ts.focal_ce_dlogits(torch.from_numpy(z["logits"]), torch.from_numpy(z["labels"]),
                              torch.from_numpy(z["weights"]), gamma, bool(rw))
    assert np.allclose(ana.numpy(), z["dlogits"], rtol=1e-4, atol=1e-7)


def test_label_mask_random_equivalence():
    rng = np.random.default_rng(0)
    for _ in range(50):
        ids = rng.integers(0, 12, size=(4, 40))
        a = ts.label_mask_loop(ids, 8, 9, 10, 11)
        b = ts.label_mask(ids, 8, 9, 10, 11)
        assert np.array_equal(a, b)


def test_vit_vs_intree_clip():
    z = np.load(os.path.join(G, "clip_tiny.npz"))
    sd = _sd(z)
    m = ovit.VisionTransformer(image_size=32, patch_size=8, width=64, layers=2, heads=4, mlp_dim=128, output_dim=8)
    e = "vision_model.embeddings."
    new = {"conv1.weight": sd[e + "patch_embedding.weight"], "class_embedding": sd[e + "class_embedding"],
           "positional_embedding": sd[e + "position_embedding.weight"],
           "ln_pre.weight": sd["vision_model.pre_layrnorm.weight"], "ln_pre.bias": sd["vision_model.pre_layrnorm.bias"],
           "ln_post.weight": sd["vision_model.post_layernorm.weight"], "ln_post.bias": sd["vision_model.post_layernorm.bias"],
           "proj": torch.eye(64)[:, :8]}
    for i in range(2):
        s, d = f"vision_model.encoder.layers.{i}.", f"transformer.resblocks.{i}."
        new[d + "attn.in_proj_weight"] = torch.cat([sd[s + f"self_attn.{n}_proj.weight"] for n in "qkv"])
        new[d + "attn.in_proj_bias"] = torch.cat([sd[s + f"self_attn.{n}_proj.bias"] for n in "qkv"])
        for a, b in [("attn.out_proj", "self_attn.out_proj"), ("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"),
                     ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")]:
            for w in ("weight", "bias"):
                new[d + a + "." + w] = sd[s + b + "." + w]
    m.load_state_dict(new)
    with torch.no_grad():
        pooled, tokens = m(torch.from_numpy(z["pixels"]))
    assert np.allclose(tokens.numpy(), z["last_hidden_state"][:, 1:], rtol=1e-4, atol=1e-4)
    assert np.allclose(pooled.numpy()[:, :8], z["pooler_output"][:, :8], rtol=1e-4, atol=1e-4)


def test_llama_vs_intree():
    z = np.load(os.path.join(G, "llama_tiny.npz"))
    m = ollama.LlamaForCausalLM(97, 64, 2, 4, 112, 1e-6)
    m.load_state_dict(_sd(z))
    ids = torch.from_numpy(z["ids"])
    loss, logits = m(ids, ids)
    assert np.allclose(logits.detach().numpy(), z["logits"], rtol=1e-4, atol=1e-4)
    assert abs(float(loss) - float(z["loss"])) < 1e-5
    loss.backward()
    for n, p in m.named_parameters():
        assert np.allclose(p.grad.numpy(), z["grad." + n], rtol=1e-3, atol=1e-5), n


@pytest.mark.parametrize("par", [0, 1])
def test_neox_vs_transformers(par):
    z = np.load(os.path.join(G, f"neox_tiny_par{par}.npz"))
    c = olm.NeoXConfig(vocab_size=128, hidden_size=80, num_hidden_layers=2, num_attention_heads=4,
                       intermediate_size=160, rotary_pct=0.5 if par else 1.0, use_parallel_residual=bool(par))
    m = olm.GPTNeoXForCausalLM(c)
    m.load_state_dict(_sd(z))
    with torch.no_grad():
        out = m(torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"]))
    valid = z["mask"].astype(bool)
    assert np.allclose(out["logits"].numpy()[valid], z["logits"][valid], rtol=1e-4, atol=1e-4)
    assert np.allclose(out["logits"].numpy(), z["logits"], rtol=1e-4, atol=1e-4)   # pad rows too


def test_opt_vs_transformers():
    z = np.load(os.path.join(G, "opt_tiny.npz"))
    c = olm.OPTConfig(vocab_size=128, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, ffn_dim=128,
                      max_position_embeddings=64)
    m = olm.OPTForCausalLM(c)
    m.load_state_dict(_sd(z))
    with torch.no_grad():
        out = m(torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"]))
    assert np.allclose(out["logits"].numpy(), z["logits"], rtol=1e-4, atol=1e-4)


def test_mpt_vs_transformers():
    from oracle import mpt as ompt
    for heads in (4, 6):
        z = np.load(os.path.join(G, f"mpt_tiny_h{heads}.npz"))
        m = ompt.MptForCausalLM(ompt.MPTConfig(vocab_size=128, d_model=16 * heads, n_layers=2, n_heads=heads, max_seq_len=64))
        missing, unexpected = m.load_state_dict(_sd(z), strict=False)
        assert not unexpected and all("lm_head" in k for k in missing)
        with torch.no_grad():
            out = m(torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"]))
        valid = torch.from_numpy(z["mask"]).bool()
        assert np.allclose(out["logits"][valid].numpy(), z["logits"][valid.numpy()], rtol=1e-4, atol=1e-4)


def test_checkpoint_key_set_matches_reference_get_checkpoint(golden_dir):
    """tests/golden/checkpoint_keys.npz = the reference's own get_checkpoint (train_utils.py:258-265) on the oracle Flamingo
    (oracle/make_golden.py ckpt).  The product's get_checkpoint is pure host logic: same keys on the same module tree."""
    import os
    import numpy as np
    import _parity as P
    from unimp_amd.train import get_checkpoint
    z = np.load(os.path.join(golden_dir, "checkpoint_keys.npz"))
    om, layout = P.build_oracle(P.TINY)
    sd = get_checkpoint(om)
    assert sorted(sd) == [str(k) for k in z["keys"]]
    assert any(k.startswith("lang_encoder.old_decoder_blocks.") for k in sd)        # the duplicate-path quirk
    assert not any(k.startswith("vision_encoder.") for k in sd)
    assert torch.equal(sd["perceiver.latents"], torch.from_numpy(z["t.perceiver.latents"]))
