"""End-to-end parity on MI355X: the HIP Flamingo train step vs the fp32 CPU oracle on identical weights / batch.

Tolerance (written here per BASELINE north_star): the HIP path keeps activations in bf16 (8-bit mantissa) between
kernels with fp32 accumulation inside them; against the fp32 oracle we require
  logits   relative L2 error <= 1e-2   (per-element bf16 rounding is 4e-3; north_star's 1e-3 is met on the loss)
  loss     relative error    <= 2e-3
  argmax   identical wherever the oracle's top-2 margin exceeds 2% of the logit scale
  grads    relative L2 error per parameter tensor <= max(3e-2, 5 x the bf16 noise floor of that tensor), the noise
           floor being the deviation of the SAME fp32 oracle re-run under bf16 autocast (tiny, ill-conditioned towers
           amplify bf16 rounding: e.g. 5% on the 128-wide OPT case even for the reference arithmetic itself)
  labels   bit-exact (integer work)
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
bf16 = torch.bfloat16


@pytest.fixture(scope="module")
def P():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import _parity
    return _parity


@pytest.mark.parametrize("cfgname", ["TINY", "TINY_OPT", "TINY_PAR", "TINY_MPT", "TINY_MOSAIC"])
def test_forward_backward_parity(P, cfgname):
    from unimp_amd.train import Trainer
    cfg = getattr(P, cfgname)
    om, layout = P.build_oracle(cfg)
    hm = P.build_hip(cfg, om, layout)
    batch = P.make_batch(cfg, layout)
    want_logits, want_loss, want_labels, want_grads = P.oracle_step(om, layout, batch)
    tr = Trainer(hm, layout.special(), lr=1e-3, gamma=2.0, use_reweight=True)
    dev = {k: v.cuda() for k, v in batch.items()}
    hm.train()
    loss, stats, out, labels = tr.forward_loss(dev)
    assert torch.equal(labels.cpu(), want_labels)
    got_logits = out["logits"].float().cpu()
    assert got_logits.shape == want_logits.shape
    e = P.rel_l2(got_logits, want_logits)
    assert e <= 1e-2, f"logits rel L2 {e}"
    assert abs(loss.item() - want_loss.item()) <= 1e-3 * abs(want_loss.item()), (loss.item(), want_loss.item())     # north_star's bound (measured 1e-5 ... 3e-4)
    top2 = want_logits.topk(2, -1).values
    sure = (top2[..., 0] - top2[..., 1]) > 0.02 * want_logits.abs().max()
    assert sure.any()
    assert torch.equal(got_logits.argmax(-1)[sure], want_logits.argmax(-1)[sure])
    ag = P.argmax_agreement(got_logits, want_logits, batch["attention_mask"].bool())     # "argmax bit-exact" as a number
    print(f"\n[{cfgname}] argmax agreement {ag['rate']:.4f} over {ag['n']} valid positions; {ag['n_sure']} with HIP margin > 8 sigma "
          f"({ag['sigma']:.2e}): identical = {ag['sure_equal']}")
    assert ag["sure_equal"] and ag["rate"] >= 0.9, ag
    tr._backward(loss)              # the production backward: weight-gradient GEMMs add straight into the flat gradient buffer
    noise = P.bf16_noise_floor(om, layout, batch, want_labels, want_grads)
    named = dict(hm.named_parameters())
    checked = 0
    for n, g in want_grads.items():
        p = named[n]
        assert p.grad is not None, n
        if g.abs().max() == 0:
            assert p.grad.float().abs().max() == 0, n
            continue
        e = P.rel_l2(p.grad, g)
        # scalar tanh-gate gradients are one heavily cancelling dot product: a single noise sample is not a bound for
        # them (ReLU towers: up to ~30 % for the reference arithmetic itself); they are checked tightly in TINY / TINY_PAR
        tol = max(3e-2, 5 * noise[n]) if g.numel() > 1 or cfgname != "TINY_OPT" else 0.5
        assert e <= tol, f"grad {n}: rel L2 {e} (bf16 noise floor {noise[n]})"
        checked += 1
    assert checked >= 20
    for n, p in named.items():
        if not p.requires_grad:
            assert n not in want_grads


@pytest.mark.parametrize("cfgname", ["TINY", "TINY_MPT"])
def test_step_results_do_not_depend_on_memory_the_step_did_not_write(P, cfgname):
    """Two optimizer steps of a tiny tower from the same weights and batch, three times in one process: as is, with every free block of the caching
    allocator filled with 0xFF (NaN in bf16 and fp32) before the steps, and with 0x00 -- whatever torch.empty() hands the kernels then holds that
    pattern.  Losses and updated parameters have the same bits in all three: no kernel of forward, loss, backward, clip or AdamW reads a byte it (or a
    kernel before it) did not write.  (tools/check_uninitialised_reads.py runs the same on the cfg2 cached decode.)"""
    from unimp_amd.train import Trainer
    if not hasattr(P, cfgname):
        pytest.skip(f"no {cfgname} config")
    cfg = getattr(P, cfgname)
    om, layout = P.build_oracle(cfg)
    batch = {k: v.cuda() for k, v in P.make_batch(cfg, layout, seed=11).items()}

    def poison(byte):
        torch.cuda.synchronize()
        sizes = [b["size"] for seg in torch.cuda.memory_snapshot() for b in seg["blocks"] if b["state"] == "inactive"]
        held = []
        for sz in sorted(sizes, reverse=True):
            try:
                t = torch.empty(sz, dtype=torch.uint8, device="cuda")
                t.fill_(byte)
                held.append(t)
            except RuntimeError:
                pass
        n = sum(t.numel() for t in held)
        del held
        torch.cuda.synchronize()
        return n

    outs = []
    for byte in (None, 0xFF, 0x00):
        tr = Trainer(P.build_hip(cfg, om, layout), layout.special(), lr=1e-3, lr_scheduler="constant")
        if byte is not None:
            assert poison(byte) > 0
        l1, _ = tr.step(batch)
        l2, _ = tr.step(batch)
        torch.cuda.synchronize()
        outs.append((l1.item(), l2.item(), tr.opt.flat_p.clone()))
        del tr
    for o in outs[1:]:
        assert o[0] == outs[0][0] and o[1] == outs[0][1], (o[:2], outs[0][:2])
        assert torch.equal(o[2], outs[0][2]), int((o[2] != outs[0][2]).sum())


def test_frozen_weight_transposed_copy_matches_default(P, monkeypatch):
    """opt-in UNIMP_FROZEN_WT: the forward MLP GEMMs of frozen towers read a cached W^T.  Same logits / loss / gradients as the
    default layout up to the summation order inside the MFMA, and the copy follows an in-place weight update."""
    from unimp_amd import functional as F_
    from unimp_amd.train import Trainer
    cfg = P.TINY
    om, layout = P.build_oracle(cfg)
    batch = {k: v.cuda() for k, v in P.make_batch(cfg, layout).items()}
    res = {}
    for flag in (False, True):
        monkeypatch.setattr(F_, "FROZEN_WT", flag)
        hm = P.build_hip(cfg, om, layout)
        tr = Trainer(hm, layout.special(), lr=1e-3, gamma=2.0, use_reweight=True)
        hm.train()
        loss, stats, out, labels = tr.forward_loss(batch)
        loss.backward()
        res[flag] = (out["logits"].float(), loss.item(), {n: p.grad.float().clone() for n, p in hm.named_parameters() if p.grad is not None})
        if flag:                                     # a frozen weight rewritten in place: the cached copy must be rebuilt
            frozen = [p for n, p in hm.named_parameters() if not p.requires_grad and hasattr(p, "_unimp_wt")]
            assert frozen, "no frozen MLP weight took the transposed path"
            with torch.no_grad():
                for p_ in frozen:
                    p_.mul_(0.5)
                again = hm(batch["vision_x"], batch["lang_x"], batch["attention_mask"])["logits"].float()
            monkeypatch.setattr(F_, "FROZEN_WT", False)
            with torch.no_grad():
                ref = hm(batch["vision_x"], batch["lang_x"], batch["attention_mask"])["logits"].float()
            assert P.rel_l2(again, ref) <= 5e-3 and P.rel_l2(again, res[True][0]) > 1e-2
    assert P.rel_l2(res[True][0], res[False][0]) <= 5e-3
    assert abs(res[True][1] - res[False][1]) <= 2e-3 * abs(res[False][1])
    for n, g in res[False][2].items():
        if g.abs().max() > 0:
            assert P.rel_l2(res[True][2][n], g) <= 3e-2, n


def test_train_steps_match_oracle_adamw(P):
    """3 optimizer steps (clip 1.0 + AdamW with the reference's decay grouping) vs the oracle's update rule."""
    from unimp_amd.train import Trainer
    from unimp_amd.optim import apply_decay
    from oracle import train_step as ots
    cfg = P.TINY
    om, layout = P.build_oracle(cfg)
    hm = P.build_hip(cfg, om, layout)
    tr = Trainer(hm, layout.special(), lr=2e-3, weight_decay=0.1, gamma=2.0, lr_scheduler="constant")
    state = {n: (torch.zeros_like(p), torch.zeros_like(p)) for n, p in om.named_parameters() if p.requires_grad}
    init = {n: p.data.clone() for n, p in om.named_parameters() if p.requires_grad}
    for step in range(1, 4):
        batch = P.make_batch(cfg, layout, seed=100 + step)
        _, want_loss, _, grads = P.oracle_step(om, layout, batch)
        tot, coef = ots.clip_coef(list(grads.values()), 1.0)
        for n, p in om.named_parameters():
            if p.requires_grad:
                m, v = state[n]
                ots.adamw_step(p.data, grads[n] * coef, m, v, step, 2e-3, 0.1 if apply_decay(n) else 0.0)
        loss, _ = tr.step({k: v.cuda() for k, v in batch.items()})
        assert abs(loss.item() - want_loss.item()) <= 5e-3 * abs(want_loss.item()), (step, loss.item(), want_loss.item())
        gn = tr.opt.grad_norm().item()
        assert abs(gn - tot) <= 3e-2 * tot, (gn, tot)
    # fp32 master weights track the oracle's fp32 parameters.  Adam normalises every element's step to ~lr, so elements
    # whose gradient is at the bf16 noise level may step differently; compare the accumulated UPDATE per tensor
    # (relative L2 <= 0.3 on tensors >= 4096 elements) and bound every element by the 3-step Adam travel 3*lr(1+wd).
    worst = 0.0
    for n, p, o, k in tr.opt.layout:
        want = dict(om.named_parameters())[n].data.reshape(-1)
        got = tr.opt.master[o:o + k].cpu()
        assert (got - want).abs().max() <= 2 * 3 * 2e-3 * 1.2, n
        if k >= 4096:
            e = P.rel_l2(got - init[n].reshape(-1), want - init[n].reshape(-1))
            worst = max(worst, e)
            assert e <= 0.3, (n, e)
    print("worst update rel-L2", worst)


def test_gate_zero_identity_and_image_locality(P):
    """architecture KATs on the HIP path: bitwise."""
    cfg = P.TINY
    om, layout = P.build_oracle(cfg, gate=0.0)
    hm = P.build_hip(cfg, om, layout).eval()
    batch = {k: v.cuda() for k, v in P.make_batch(cfg, layout).items()}
    with torch.no_grad():
        a = hm(batch["vision_x"], batch["lang_x"], batch["attention_mask"])["logits"]
        vx = batch["vision_x"].clone()
        vx[:, 1] += 1.0
        b = hm(vx, batch["lang_x"], batch["attention_mask"])["logits"]
    assert torch.equal(a, b)              # gates 0: images cannot matter
    om2, layout = P.build_oracle(cfg, gate=0.5)
    hm2 = P.build_hip(cfg, om2, layout).eval()
    with torch.no_grad():
        a = hm2(batch["vision_x"], batch["lang_x"], batch["attention_mask"])["logits"]
        b = hm2(vx, batch["lang_x"], batch["attention_mask"])["logits"]
    ids = batch["lang_x"]
    for r in range(ids.shape[0]):
        second = (ids[r] == layout.media).nonzero()[1].item()
        assert torch.equal(a[r, :second], b[r, :second])
        assert not torch.equal(a[r, second:], b[r, second:])


def test_llama_tower_vs_intree_reference(P, golden_dir):
    """§8 a-9: RMSNorm / RoPE / SwiGLU / causal attention chain against the in-tree UniMP/xformers_model/llama.py
    (fixture tests/golden/llama_hd64.npz captured from the reference itself; bf16-representable weights)."""
    import os
    import numpy as np
    from unimp_amd.lm import LlamaForCausalLM, LlamaConfig
    z = np.load(os.path.join(golden_dir, "llama_hd64.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
    with torch.device("cuda"):
        m = LlamaForCausalLM(LlamaConfig(vocab_size=200, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                                         num_attention_heads=2, max_position_embeddings=64))
    m.to(dtype=bf16)
    missing, unexpected = m.load_state_dict(sd, strict=True)
    m.requires_grad_(False)
    m.model.embed_tokens.weight.requires_grad_(True)
    m.lm_head.weight.requires_grad_(True)
    ids = torch.from_numpy(z["ids"]).cuda()
    out = m(ids, labels=ids)
    e = P.rel_l2(out["logits"], torch.from_numpy(z["logits"]))
    assert e <= 1e-2, f"logits rel L2 {e}"
    assert abs(out[0].item() - float(z["loss"])) <= 2e-3 * float(z["loss"]), (out[0].item(), float(z["loss"]))
    out[0].backward()
    for n in ("model.embed_tokens.weight", "lm_head.weight"):       # embedding grad = the whole dX chain of the tower
        g = dict(m.named_parameters())[n].grad
        e = P.rel_l2(g, torch.from_numpy(z["grad." + n]))
        assert e <= 3e-2, f"grad {n}: rel L2 {e}"


def test_vit_vs_intree_clip_reference(P, golden_dir):
    """§8 a-3/4/5: the HIP ViT (conv1-as-GEMM, fused in_proj, QuickGELU MLP, pre-LN) against the in-tree
    UniMP/xformers_model/clip.py (fixture tests/golden/clip_hd64.npz captured from the reference)."""
    import os
    import numpy as np
    from unimp_amd.vit import VisionTransformer
    z = np.load(os.path.join(golden_dir, "clip_hd64.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
    e_ = "vision_model.embeddings."
    new = {"conv1.weight": sd[e_ + "patch_embedding.weight"], "class_embedding": sd[e_ + "class_embedding"],
           "positional_embedding": sd[e_ + "position_embedding.weight"],
           "ln_pre.weight": sd["vision_model.pre_layrnorm.weight"], "ln_pre.bias": sd["vision_model.pre_layrnorm.bias"],
           "ln_post.weight": sd["vision_model.post_layernorm.weight"], "ln_post.bias": sd["vision_model.post_layernorm.bias"],
           "proj": torch.eye(128)[:, :8].contiguous()}
    for i in range(2):
        s_, d_ = f"vision_model.encoder.layers.{i}.", f"transformer.resblocks.{i}."
        new[d_ + "attn.in_proj_weight"] = torch.cat([sd[s_ + f"self_attn.{n}_proj.weight"] for n in "qkv"])
        new[d_ + "attn.in_proj_bias"] = torch.cat([sd[s_ + f"self_attn.{n}_proj.bias"] for n in "qkv"])
        for a, b in [("attn.out_proj", "self_attn.out_proj"), ("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"),
                     ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")]:
            for w in ("weight", "bias"):
                new[d_ + a + "." + w] = sd[s_ + b + "." + w]
    with torch.device("cuda"):
        v = VisionTransformer(image_size=32, patch_size=8, width=128, layers=2, heads=2, mlp_dim=256, output_dim=8)
    v.to(dtype=bf16)
    v.load_state_dict(new)
    v.output_tokens = True
    v.requires_grad_(False)
    with torch.no_grad():
        pooled, tokens = v(torch.from_numpy(z["pixels"]).cuda())
    want = torch.from_numpy(z["last_hidden_state"])
    e = P.rel_l2(tokens, want[:, 1:])
    assert e <= 1e-2, f"tokens rel L2 {e}"
    e = P.rel_l2(pooled, torch.from_numpy(z["pooler_output"])[:, :8])
    assert e <= 1e-2, f"pooled rel L2 {e}"


def test_sparse_head_step_equals_dense_step():
    """Trainer(sparse_head=True) -- head + focal loss on the labeled rows only -- gives the dense step's loss and update
    (unlabeled rows contribute nothing to either; the GEMMs just see fewer rows, so agreement is at bf16 rounding level)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import _parity as P
    from unimp_amd.train import Trainer
    cfg = P.TINY
    om, layout = P.build_oracle(cfg)
    batch = {k: v.cuda() for k, v in P.make_batch(cfg, layout).items()}
    batch["vision_x"] = batch["vision_x"].to(torch.bfloat16)
    res = {}
    for sparse in (False, True):
        hm = P.build_hip(cfg, om, layout)
        tr = Trainer(hm, layout.special(), lr=1e-2, lr_scheduler="constant", sparse_head=sparse)
        before = tr.opt.master.clone()
        loss, stats = tr.step(batch)
        res[sparse] = (float(loss), stats.float().cpu(), (tr.opt.master - before).cpu())
    (l0, s0, d0), (l1, s1, d1) = res[False], res[True]
    assert abs(l0 - l1) <= 2e-3 * abs(l0), (l0, l1)
    assert float(s0[1]) == float(s1[1]) > 0                               # same number of labeled tokens
    rel = float((d0 - d1).norm() / d0.norm())
    assert rel < 0.1, rel                                                 # Adam's sign-like first step amplifies rounding: compare directions
    cos = float((d0 * d1).sum() / (d0.norm() * d1.norm()))
    assert cos > 0.99, cos


@pytest.mark.parametrize("n_heads,d_model", [(2, 128), (6, 384)])
def test_mpt_tower_vs_transformers(n_heads, d_model):
    """The MPT tower (OpenFlamingo-9B's language model: ALiBi, no biases, tied head) against the installed transformers'
    MptForCausalLM on identical bf16-representable weights: logits, the tied-embedding gradient of a CE loss, and the KV-cached
    decode.  6 heads exercise the interleaved slopes of a non-power-of-two head count."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    transformers = pytest.importorskip("transformers")
    from transformers import MptConfig, MptForCausalLM as HFMpt
    from unimp_amd.lm import MPTConfig, build_lm
    torch.manual_seed(0)
    V, L, B = 320, 96, 2
    hf = HFMpt(MptConfig(d_model=d_model, n_heads=n_heads, n_layers=2, expansion_ratio=4, max_seq_len=256, vocab_size=V,
                         no_bias=True, layer_norm_epsilon=1e-5)).eval()
    with torch.no_grad():
        for p_ in hf.parameters():
            p_.copy_((p_ * (3.0 if p_.dim() >= 2 else 1.0)).to(torch.bfloat16).float())
    mine = build_lm(MPTConfig(vocab_size=V, d_model=d_model, n_layers=2, n_heads=n_heads)).to(device="cuda", dtype=torch.bfloat16)
    missing, unexpected = mine.load_state_dict(hf.state_dict(), strict=False)
    assert not [k for k in missing if "lm_head" not in k] and not [k for k in unexpected if "lm_head" not in k], (missing, unexpected)
    ids = torch.randint(0, V, (B, L))
    mask = torch.ones(B, L, dtype=torch.long); mask[1, 70:] = 0
    labels = ids.clone(); labels[mask == 0] = -100
    out = hf(input_ids=ids, attention_mask=mask, labels=labels)
    out.loss.backward()
    got = mine(ids.cuda(), mask.cuda(), labels=labels.cuda())
    valid = mask.bool()
    rel = float((got["logits"].float().cpu()[valid] - out.logits[valid]).norm() / out.logits[valid].norm())
    assert rel <= 1e-2, rel
    assert abs(float(got[0]) - float(out.loss)) <= 2e-3 * float(out.loss)
    got[0].backward()
    gw, ww = mine.transformer.wte.weight.grad.float().cpu(), hf.transformer.wte.weight.grad
    assert float((gw - ww).norm() / ww.norm()) <= 3e-2
    # cached decode: prefill on the first 60 tokens, then 3 single-token steps == full forward
    with torch.no_grad():
        full = mine(ids[:1, :63].cuda())["logits"].float()
        o = mine(ids[:1, :60].cuda(), use_cache=True)
        steps = [o["logits"][:, -1].float()]
        for j in range(60, 63):
            o = mine(ids[:1, j:j + 1].cuda(), past_key_values=o.past_key_values, use_cache=True)
            steps.append(o["logits"][:, -1].float())
    for j, lg in enumerate(steps):
        assert float((lg - full[:, 59 + j]).abs().max()) <= 2e-2 * float(full.abs().max()), j


def test_flamingo_over_mpt_generate_cached_equals_rescoring():
    """Flamingo over an MPT tower end to end: the KV-cached / HIP-graph decode (ALiBi through the decode step, tied head)
    returns the tokens of full re-scoring."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from unimp_amd import create_model_and_transforms
    from unimp_amd.factory import SyntheticTokenizer
    from unimp_amd.lm import MPTConfig
    from unimp_amd.synthetic import TokenLayout, make_batch
    torch.manual_seed(1)
    tok = SyntheticTokenizer(base_vocab=300)
    model, _, tok = create_model_and_transforms(
        dict(image_size=32, patch_size=8, width=128, layers=1, heads=2, mlp_dim=256, output_dim=64), None,
        MPTConfig(vocab_size=400, d_model=256, n_layers=2, n_heads=4), None, cross_attn_every_n_layers=1, tokenizer=tok, device="cuda")
    layout = TokenLayout(300, 40, 16)
    model.lang_encoder.resize_token_embeddings(layout.vocab)
    model.media_token_id = model.lang_encoder.media_token_id = layout.media
    model.eoc_token_id = layout.eoc
    with torch.no_grad():
        for n, p_ in model.named_parameters():
            if p_.dim() >= 2 and "latents" not in n:
                p_.normal_(0, 0.08)
        for g_ in model.lang_encoder.gated_cross_attn_layers:
            g_.attn_gate.fill_(0.5); g_.ff_gate.fill_(0.5)
    bt = make_batch(layout, 1, 2, 40, image_size=32, seed=3, device="cuda", vision_dtype=torch.bfloat16)
    n = int(bt["attention_mask"][0].sum()) - 2
    ids, vx = bt["lang_x"][:, :n], bt["vision_x"]
    kw = dict(max_new_tokens=6, eos_token_id=layout.eos, pad_token_id=layout.eos)
    a = model.generate(vx, ids, use_cache=True, **kw)
    b = model.generate(vx, ids, use_cache=False, **kw)
    m = min(a.shape[1], b.shape[1])
    assert int((a[0, :m] == b[0, :m]).long().cumprod(0).sum()) >= n + 2, (a.tolist(), b.tolist())
    c = model.generate(vx, ids, use_cache=True, use_graph=False, **kw)
    assert torch.equal(a, c)
    k = model.generate(vx, ids, num_beams=3, num_return_sequences=3, early_stopping=True, **kw)
    assert k.shape[0] == 3 and torch.equal(k[:, :n], ids.expand(3, -1))


@pytest.mark.parametrize("fuse", [None, False])
def test_gradient_accumulation_equals_one_step_on_the_same_batch(fuse):
    """grad_accum = 2 over two identical micro-batches is one optimizer step with that batch's gradient ((g + g) / 2), in the default
    (fused: both micro-batches in one pass) and in the sequential form."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import _parity as P
    from unimp_amd.train import Trainer
    cfg = P.TINY
    om, layout = P.build_oracle(cfg)
    batch = {k: v.cuda() for k, v in P.make_batch(cfg, layout).items()}
    res = {}
    for ga in (1, 2):
        hm = P.build_hip(cfg, om, layout)
        tr = Trainer(hm, layout.special(), lr=1e-2, lr_scheduler="constant", grad_accum=ga, fuse_accum=fuse)
        before = tr.opt.master.clone()
        for _ in range(ga):
            tr.step(batch)
            if ga == 2 and _ == 0:
                assert torch.equal(tr.opt.master, before) and tr.sched_step == 0        # no update after the first micro-batch
        res[ga] = (tr.opt.master - before).cpu()
        assert tr.sched_step == 1 and tr.opt.step_count == 1
    cos = float((res[1] * res[2]).sum() / (res[1].norm() * res[2].norm()))
    assert cos > 0.995 and float((res[1] - res[2]).norm() / res[1].norm()) < 0.1, cos


def test_scheduler_and_accumulation_stepping(P):
    """SURVEY Appendix B.6.  mmrec.py:255 calls ``lr_scheduler.step()`` after EVERY micro-batch, but sizes the cosine
    schedule in OPTIMIZER steps (warmup // GA, total // GA: mmrec.py:687-693), and under the shipped DeepSpeed ZeRO-2 launch
    (unimp_task.sh:9, accelerate_config_zero2.yaml) that call is a no-op: accelerate's DeepSpeed wrapper steps the engine --
    clip, optimizer, scheduler -- once per gradient-accumulation boundary (accelerate utils/deepspeed.py
    DeepSpeedEngineWrapper.backward).  The build follows the shipped behaviour: with grad_accum = GA the learning rate is
    the k-th value of transformers' cosine schedule built with (warmup // GA, total // GA) for ALL micro-batches of
    optimizer step k, one optimizer step and one scheduler step per GA micro-batches, gradients summed over them and
    scaled by 1 / GA.  (A literal non-DeepSpeed reading of mmrec.py:255 would advance the schedule GA times faster.)"""
    import transformers
    from unimp_amd.train import Trainer
    cfg = P.TINY
    om, layout = P.build_oracle(cfg)
    hm = P.build_hip(cfg, om, layout)
    GA, warm, total, base = 2, 4, 24, 3e-4
    tr = Trainer(hm, layout.special(), lr=base, lr_scheduler="cosine", warmup_steps=warm // GA, total_steps=total // GA, grad_accum=GA,
                 fuse_accum=False)                       # the sequential form: 1 / GA rides in the optimizer's gradient scale
    ref_opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=base)
    ref = transformers.get_cosine_schedule_with_warmup(ref_opt, num_warmup_steps=warm // GA, num_training_steps=total // GA)
    used, scales = [], []
    real_step = tr.opt.step
    def spy(lr=None, grad_scale=1.0):
        used.append(lr); scales.append(grad_scale)
        return real_step(lr=lr, grad_scale=grad_scale)
    tr.opt.step = spy
    batches = [{k: v.cuda() for k, v in P.make_batch(cfg, layout, seed=900 + i).items()} for i in range(2)]
    want = []
    for micro in range(8):
        if micro % GA == 0:
            want.append(ref_opt.param_groups[0]["lr"])
        tr.step(batches[micro % 2])
        if micro % GA == GA - 1:
            ref_opt.step(); ref.step()
    assert len(used) == 8 // GA and tr.sched_step == 8 // GA and tr.opt.step_count == 8 // GA
    assert all(abs(a - b) <= 1e-12 for a, b in zip(used, want)), (used, want)
    assert all(abs(s - 1.0 / GA) < 1e-12 for s in scales)
    tr.dp.remove()


def test_compact_head_backward_equals_dense(P):
    """The default Trainer restricts the LM head's backward to the labeled positions (functional.DenseHeadLossFn): same dense
    logits (bitwise), same loss (bitwise), and every gradient equal to the dense path's up to the fp32 summation order of the
    head's weight gradient (the contraction runs over n labeled rows instead of B*L rows, B*L - n of which are zero)."""
    from unimp_amd.train import Trainer
    cfg = P.TINY
    om, layout = P.build_oracle(cfg)
    batch = {k: v.cuda() for k, v in P.make_batch(cfg, layout).items()}
    res = {}
    for dense in (True, False):
        hm = P.build_hip(cfg, om, layout)
        tr = Trainer(hm, layout.special(), lr=1e-3, gamma=2.0, use_reweight=True, dense_head_backward=dense)
        hm.train()
        loss, stats, out, labels = tr.forward_loss(batch)
        loss.backward()
        res[dense] = (out["logits"].clone(), loss.item(), stats.clone(), {n: p.grad.float().clone() for n, p in hm.named_parameters() if p.grad is not None})
        tr.dp.remove()
    assert torch.equal(res[True][0], res[False][0]) and res[True][1] == res[False][1] and torch.equal(res[True][2], res[False][2])
    assert set(res[True][3]) == set(res[False][3])
    for n, g in res[True][3].items():
        if g.abs().max() == 0:
            assert res[False][3][n].abs().max() == 0, n
        else:
            assert P.rel_l2(res[False][3][n], g) <= 4e-3, (n, P.rel_l2(res[False][3][n], g))


def test_labeled_rows_without_a_blocking_nonzero():
    """Trainer._labeled_rows_begin / _end (count read behind an event, after the forward is queued) return exactly the flat rows
    b * L + j that ``nonzero()`` on the shifted label mask gives, in the same order -- including no labeled position at all and
    a label in the last column (position L - 1 has no next token: never a scored row)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from unimp_amd.train import Trainer
    tr = Trainer.__new__(Trainer)
    tr._cnt_host = None
    g = torch.Generator().manual_seed(5)
    for B, L, frac in [(3, 17, 0.2), (8, 512, 0.02), (2, 9, 0.0), (1, 5, 1.0)]:
        labels = torch.where(torch.rand(B, L, generator=g) < frac, torch.randint(0, 100, (B, L), generator=g), torch.full((B, L), -100)).cuda()
        bj = (labels[:, 1:] != -100).nonzero()
        want = bj[:, 0] * L + bj[:, 1]
        pend = tr._labeled_rows_begin(labels)
        torch.mm(torch.randn(512, 512, device="cuda"), torch.randn(512, 512, device="cuda"))     # work queued between begin and end
        got = tr._labeled_rows_end(pend)
        assert got.dtype == torch.int64 and got.is_contiguous() and torch.equal(got, want), (B, L, frac)
        cpu = tr._labeled_rows_end(tr._labeled_rows_begin(labels.cpu()))
        assert torch.equal(cpu, want.cpu())


def test_direct_weight_gradients_equal_autograd_path(P):
    """Single rank: the dW GEMMs accumulate into the flat gradient buffer themselves (functional.WGRAD_SINK, epilogue
    ``accumulate``) instead of returning a temporary for autograd's ``.grad +=``.  Same gradients up to one bf16 rounding
    (the direct path rounds acc + grad once, autograd rounds the GEMM output and then the sum), also when two micro-batches
    accumulate; the sink is really used (every 2-D trainable weight of the gated blocks and the Perceiver), and split-K
    weight gradients (narrow projections) take the same route."""
    from unimp_amd.train import Trainer
    from unimp_amd import functional as Fn
    cfg = P.TINY
    om, layout = P.build_oracle(cfg)
    batch = {k: v.cuda() for k, v in P.make_batch(cfg, layout).items()}
    res, hits = {}, 0
    for direct in (False, True):
        hm = P.build_hip(cfg, om, layout)
        tr = Trainer(hm, layout.special(), lr=1e-3, gamma=2.0, use_reweight=True, direct_wgrad=direct)
        assert (tr._sink is not None) == direct
        hm.train()
        if direct:
            orig = tr._sink.view_of
            def counted(w, orig=orig):
                nonlocal hits
                v = orig(w)
                hits += v is not None
                return v
            tr._sink.view_of = counted
        for _ in range(2):                      # two micro-batches add up in the buffer
            loss, stats, out, labels = tr.forward_loss(batch)
            tr._backward(loss)
        assert Fn.WGRAD_SINK is None
        res[direct] = {n: p.grad.float().clone() for n, p in hm.named_parameters() if p.grad is not None}
        tr.dp.remove()
    assert hits >= 2 * 10, hits
    assert set(res[True]) == set(res[False])
    for n, g in res[False].items():
        if g.abs().max() == 0:
            assert res[True][n].abs().max() == 0, n
        else:
            assert P.rel_l2(res[True][n], g) <= 6e-3, (n, P.rel_l2(res[True][n], g))


@pytest.mark.parametrize("interleaved,rotary_pct", [(True, 1.0), (True, 0.25), (False, 1.0)])
def test_rotary_epilogue_block_equals_separate_rope_pass(monkeypatch, interleaved, rotary_pct):
    """A frozen self-attention block at the LM's head size: the QKV GEMM rotates q / k in its epilogue (permuted projection
    rows, adjacent pairs, on-the-fly cos / sin) and the attention backward rotates dq / dk back in the same layout --
    output and input gradient equal the path with the two rope_ passes up to bf16 rounding."""
    from unimp_amd import functional as Fn, ops
    nh, hd, B, L = 4, 80, 2, 256
    H, rot = nh * hd, int(80 * rotary_pct) // 8 * 8
    g = torch.Generator().manual_seed(5)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16).cuda()
    x = r(B, L, H)
    ln_w, ln_b = (1 + 0.1 * torch.randn(H, generator=g)).to(torch.bfloat16).cuda(), r(H, sc=0.1)
    wqkv, bqkv, wd, bd = r(3 * H, H, sc=H ** -0.5), r(3 * H, sc=0.1), r(H, H, sc=H ** -0.5), r(H, sc=0.1)
    inv = 1.0 / (10000.0 ** (torch.arange(0, rot, 2, dtype=torch.float32) / rot))
    fr = torch.arange(L, dtype=torch.float32)[:, None] * inv[None]
    rope = (fr.cos().contiguous().cuda(), fr.sin().contiguous().cuda(), rot, 10000.0)
    dy = r(B, L, H)
    monkeypatch.setattr(ops, "gemm_rope_variant", lambda M, N, K, b_ks, dev: None if b_ks else 4)
    res = {}
    for fused in (False, True):
        monkeypatch.setattr(Fn, "ROPE_EPILOGUE", fused)
        xr = x.clone().requires_grad_(True)
        out = Fn.self_attn_block(xr, ln_w, ln_b, wqkv, bqkv, wd, bd, nh, rope=rope, interleaved=interleaved)
        out.backward(dy)
        res[fused] = (out.detach().float(), xr.grad.float())
    assert hasattr(wqkv, "_unimp_rope_perm")                      # the fused path really ran
    for a, b_, name in zip(res[True], res[False], ("out", "dx")):
        e = float((a - b_).norm() / b_.norm())
        assert e <= 6e-3, (name, e)


def test_fp8_frozen_towers_track_the_bf16_path(P, monkeypatch):
    """F4 (BASELINE config 5: "fp8 MFMA weights"): with functional.FP8_FROZEN the frozen Linear layers of the LM and the ViT
    run on the MX-fp8 GEMM (e4m3 elements, E8M0 scale per 32 k, activations quantised on the fly, fp32 accumulate), forward
    and dX.  e4m3 has a 3-bit mantissa: per element 2^-4 relative, averaged down by the contraction.  Bounds (measured values
    are printed): loss within 2e-2 of the bf16 HIP path and of the fp32 oracle, logits rel-L2 <= 8e-2 vs bf16, every trainable
    gradient tensor rel-L2 <= 0.25 vs bf16, the scalar gates <= 0.6 (gradients flow back through the quantised frozen tower)."""
    from unimp_amd import functional as F_, ops
    from unimp_amd.train import Trainer
    cfg = P.TINY_MX
    monkeypatch.setattr(F_, "FP8_MIN_DIM", 128)          # the toy widths (128 / 256) on the MX kernel: LM and ViT (default 2048: LM widths only)
    om, layout = P.build_oracle(cfg)
    batch = P.make_batch(cfg, layout)
    _, want_loss, _, _ = P.oracle_step(om, layout, batch)
    dev = {k: v.cuda() for k, v in batch.items()}
    calls = []
    real = ops.gemm_mx
    monkeypatch.setattr(ops, "gemm_mx", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    res = {}
    for flag in (False, True):
        monkeypatch.setattr(F_, "FP8_FROZEN", flag)
        hm = P.build_hip(cfg, om, layout)
        tr = Trainer(hm, layout.special(), lr=1e-3, gamma=2.0)
        hm.train()
        n0 = len(calls)
        loss, stats, out, labels = tr.forward_loss(dev)
        loss.backward()
        res[flag] = (out["logits"].float().cpu(), loss.item(), {n: p.grad.float().cpu().clone() for n, p in hm.named_parameters() if p.grad is not None},
                     len(calls) - n0)
        tr.dp.remove()
    assert res[False][3] == 0 and res[True][3] >= 2 * (4 + 4) + 2 * 4, res[True][3]      # LM fwd + dX (4 GEMMs each per layer) + ViT blocks
    e_log = P.rel_l2(res[True][0], res[False][0])
    e_loss = abs(res[True][1] - res[False][1]) / abs(res[False][1])
    e_or = abs(res[True][1] - want_loss.item()) / abs(want_loss.item())
    worst = max((P.rel_l2(res[True][2][n], g), n) for n, g in res[False][2].items() if g.abs().max() > 0 and g.numel() > 1)
    worst1 = max((P.rel_l2(res[True][2][n], g), n) for n, g in res[False][2].items() if g.abs().max() > 0 and g.numel() == 1)
    print(f"\n[fp8 frozen towers] logits rel-L2 vs bf16 {e_log:.3e}; loss {res[True][1]:.5f} vs bf16 {res[False][1]:.5f} ({e_loss:.2e}) vs fp32 oracle "
          f"{want_loss.item():.5f} ({e_or:.2e}); worst gradient rel-L2 vs bf16 {worst[0]:.3e} ({worst[1]}), scalar gates {worst1[0]:.3e} "
          f"({worst1[1]}); {res[True][3]} MX GEMMs")
    # the scalar tanh-gate gradients are single, heavily cancelling dot products (see test_forward_backward_parity): looser bound
    assert e_log <= 8e-2 and e_loss <= 2e-2 and e_or <= 2e-2 and worst[0] <= 0.25 and worst1[0] <= 0.6


def test_mask_lm_head_keeps_only_the_answer_row(P):
    """mmrec.py:218-229 (--mask_lm_head): after backward the embedding / head gradients are multiplied by a mask that is 1 on the
    <answer> token's row only.  Trainer(mask_lm_head=True): after an optimizer step every other row of the input embedding and
    of the head is unchanged (AdamW without weight decay on these tensors: a zero gradient with zero moments moves nothing),
    the <answer> row moved."""
    from unimp_amd.train import Trainer
    cfg = P.TINY
    om, layout = P.build_oracle(cfg)
    hm = P.build_hip(cfg, om, layout)
    emb, head = hm.lang_encoder.get_input_embeddings().weight, hm.lang_encoder.get_output_embeddings().weight
    e0, h0 = emb.detach().clone(), head.detach().clone()
    tr = Trainer(hm, layout.special(), lr=1e-2, mask_lm_head=True)
    tr.step({k: v.cuda() for k, v in P.make_batch(cfg, layout).items()})
    a = layout.answer
    for w, w0 in ((emb, e0), (head, h0)):
        d = (w.detach().float() - w0.float()).abs().sum(1)
        assert d[a] > 0, "the <answer> row must train"
        d[a] = 0
        assert float(d.max()) == 0.0, "rows other than <answer> must not move"
    tr.dp.remove()


def test_precision_choice_is_bf16_not_the_reference_default(P):
    """the reference's default --precision amp is torch.cuda.amp fp16 autocast (train_utils.py:17-19)
    layered on a DeepSpeed bf16 engine; every shipped script keeps that default.  BASELINE names bf16, and this build computes in
    bf16 throughout (parameters, activations, logits) with fp32 accumulation / statistics / master weights -- asserted here so
    the deviation is explicit."""
    from unimp_amd.train import Trainer
    cfg = P.TINY
    om, layout = P.build_oracle(cfg)
    hm = P.build_hip(cfg, om, layout)
    assert all(p.dtype == torch.bfloat16 for p in hm.parameters())
    tr = Trainer(hm, layout.special())
    loss, stats, out, _ = tr.forward_loss({k: v.cuda() for k, v in P.make_batch(cfg, layout).items()})
    assert out["logits"].dtype == torch.bfloat16 and loss.dtype == torch.float32 and tr.opt.master.dtype == torch.float32
    tr.dp.remove()


@pytest.mark.parametrize("ga", [1, 2])
def test_graphed_micro_step_equals_eager(P, ga):
    """Trainer(graph=True): the forward + loss + backward of a micro-batch replayed as one HIP graph (first micro-step eager, second
    captured, then replays; mmrec.py's --gradient_accumulation_steps on top) must give the bits of the same steps launched kernel
    by kernel: same losses, same fp32 master weights after every optimizer step, fresh data each micro-step."""
    from unimp_amd.train import Trainer
    cfg = P.TINY
    om, layout = P.build_oracle(cfg)
    batches = [{k: v.cuda() for k, v in P.make_batch(cfg, layout, seed=700 + i).items()} for i in range(6 * ga)]
    res = {}
    for mode in ("eager", "graph"):
        hm = P.build_hip(cfg, om, layout)
        tr = Trainer(hm, layout.special(), lr=1e-3, lr_scheduler="constant", grad_accum=ga, dense_head_backward=True, graph=mode == "graph",
                     fuse_accum=False)        # both modes run the micro-steps one by one (the eager default would fuse them into one pass)
        losses, masters = [], []
        for i, b in enumerate(batches):
            loss, stats = tr.step(b)
            losses.append(loss.item())
            if (i + 1) % ga == 0:
                masters.append(tr.opt.master.clone())
        res[mode] = (losses, masters)
        if mode == "graph":
            assert tr._graph is not None and tr._graph["key"] is not None, "no graph was captured"
            # a batch of another shape falls back to an eager warm-up, then gets its own graph
            other = {k: v[:1].contiguous() for k, v in batches[0].items()}
            for _ in range(3 * ga):
                tr.step(other)
            assert tr._graph["static"]["lang_x"].shape[0] == 1
    assert res["eager"][0] == res["graph"][0], (res["eager"][0], res["graph"][0])
    assert all(torch.equal(a, b) for a, b in zip(res["eager"][1], res["graph"][1]))
    assert len(set(res["eager"][0])) == len(batches)                  # the replays really consumed fresh data


def test_fp8_loss_curve_tracks_bf16(P, monkeypatch):
    """cfg5's training dynamics with fp8 frozen towers: 48 optimizer steps on the same 8 batches (cycled), bf16 HIP against fp8 HIP
    from identical initial weights.  The trainable blocks stay bf16 in both; the quantised frozen towers perturb activations and
    the gradients that flow back through them by a few per cent per step.  Asserted: both curves fall, the fp8 curve stays within
    10 % of the bf16 curve at every step (measured: <= 2.6 %), and the mean loss of the last 8 steps agrees within 5 % (measured 0.2 %).
    Round 4: the run stops at step 48.  At this learning rate the toy model's training goes unstable near step 58 in BOTH precisions
    (bf16: 3.70 then 21.70 at steps 58 / 59, fp8: 29.97 / 22.29 -- tools/fp8curve.py); which step the spike lands on moves with
    any bit-level change, and a spike one step apart read as a 7 x "gap" under the old 60-step bound.  That is the optimizer's chaos,
    not the fp8 path; the cfg5-width run against a measured chaos floor is test_widths_gpu.py::test_cfg5_width_fp8_loss_curve_...."""
    from unimp_amd import functional as F_
    from unimp_amd.train import Trainer
    cfg = P.TINY_MX
    monkeypatch.setattr(F_, "FP8_MIN_DIM", 128)
    om, layout = P.build_oracle(cfg)
    batches = [{k: v.cuda() for k, v in P.make_batch(cfg, layout, seed=900 + i).items()} for i in range(8)]
    curves = {}
    for flag in (False, True):
        monkeypatch.setattr(F_, "FP8_FROZEN", flag)
        hm = P.build_hip(cfg, om, layout)
        tr = Trainer(hm, layout.special(), lr=2e-3, lr_scheduler="constant", gamma=2.0)
        curves[flag] = [tr.step(batches[i % 8])[0].item() for i in range(48)]
        tr.dp.remove()
    a, b = curves[False], curves[True]
    dev = max(abs(x - y) / abs(x) for x, y in zip(a, b))
    tail_a, tail_b = sum(a[-8:]) / 8, sum(b[-8:]) / 8
    print(f"\\n[fp8 loss curve] bf16 {a[0]:.3f} -> {tail_a:.3f}; fp8 {b[0]:.3f} -> {tail_b:.3f}; max relative gap over 48 steps {dev:.3e}")
    assert tail_a < 0.7 * a[0] and tail_b < 0.7 * b[0]
    assert dev <= 0.10 and abs(tail_a - tail_b) <= 0.05 * tail_a


@pytest.mark.parametrize("reweight", [True, False])
def test_fused_accumulation_equals_sequential(P, reweight):
    """Trainer(grad_accum=2, fuse_accum=True): the two micro-batches of an optimizer step in ONE pass, with the reference's
    per-micro-batch loss normalisation (mmrec.py:213 + accelerate's 1/GA) carried by per-sample weights.  Against the sequential
    accumulation on micro-batches with DIFFERENT numbers of labeled positions and different padded lengths: the same loss (mean
    of the two micro-batch losses), the same summed gradient up to bf16 summation order, the same optimizer trajectory."""
    from unimp_amd.train import Trainer
    cfg = P.TINY
    om, layout = P.build_oracle(cfg)
    mbs = []
    for i in range(4):
        b = {k: v.cuda() for k, v in P.make_batch(cfg, layout, seed=810 + i).items()}
        if i % 2 == 1:                          # a shorter micro-batch with fewer samples: other label counts, other padded length
            n = int(b["attention_mask"].sum(1).max())
            b = {k: (v[:1, :n] if k in ("lang_x", "attention_mask") else v[:1]) for k, v in b.items()}
        mbs.append(b)
    res = {}
    for mode in ("sequential", "fused"):
        hm = P.build_hip(cfg, om, layout)
        tr = Trainer(hm, layout.special(), lr=1e-3, lr_scheduler="constant", grad_accum=2, fuse_accum=mode == "fused",
                     gamma=2.0 if reweight else 0.0, use_reweight=reweight)
        grads, real = [], tr.opt.step
        tr.opt.step = lambda lr=None, grad_scale=1.0: (grads.append(tr.opt.flat_g.float() * grad_scale), real(lr=lr, grad_scale=grad_scale))[1]
        losses, pend = [], []
        for i, b in enumerate(mbs):
            out = tr.step(b)
            loss, _ = out                                           # StepOut unpacks like the old (loss, stats) tuple
            losses.append(None if loss is None else loss.item())
            pend.append(out.pending)
        assert pend == ([True, False, True, False] if mode == "fused" else [False] * 4), (mode, pend)
        assert tr.flush() is None                                   # nothing buffered after complete groups
        res[mode] = (losses, grads, tr.opt.master.clone())
        tr.dp.remove()
    (ls, gs, ms), (lf, gf, mf) = res["sequential"], res["fused"]
    # fuse_accum=None is decided at the first micro-batch: fused while GA x B x L fits the token budget, sequential beyond it or under a graph
    import unimp_amd.train as T_
    auto = Trainer(P.build_hip(cfg, om, layout), layout.special(), grad_accum=2)
    assert not auto.fuse_accum and auto.step(mbs[0]).pending and auto.fuse_accum
    old = T_.FUSE_TOKEN_BUDGET
    try:
        T_.FUSE_TOKEN_BUDGET = 1
        big = Trainer(P.build_hip(cfg, om, layout), layout.special(), grad_accum=2)
        assert not big.step(mbs[0]).pending and not big.fuse_accum
    finally:
        T_.FUSE_TOKEN_BUDGET = old
    assert not Trainer(P.build_hip(cfg, om, layout), layout.special(), grad_accum=2, graph=True, dense_head_backward=True).fuse_accum
    with pytest.raises(ValueError):
        Trainer(P.build_hip(cfg, om, layout), layout.special(), grad_accum=2, graph=True, fuse_accum=True)
    assert len(gs) == len(gf) == 2
    for k in range(2):
        want = 0.5 * (ls[2 * k] + ls[2 * k + 1])                  # accelerate: each micro-batch loss / GA
        assert abs(lf[2 * k + 1] - want) <= 2e-3 * abs(want), (lf, ls)
        e = P.rel_l2(gf[k], gs[k])
        assert e <= 2e-2, (k, e)
    assert lf[0] != lf[0] and lf[2] == lf[1]                      # buffered micro-steps: NaN before the first optimizer step, then the previous step's loss
    assert P.rel_l2(mf, ms) <= 5e-3                               # fp32 masters after two AdamW steps (Adam amplifies tiny gradients' noise)


def test_overlapped_optimizer_equals_serial(P):
    """Trainer(overlap_optimizer=True): clip + AdamW on a second stream under the next step's frozen ViT forward (VERDICT r4 #5a).  Same
    kernels in the same order per buffer, the main stream waits for the update before the Perceiver reads a trainable parameter: after
    five steps on changing batches the losses, the bf16 parameters and the fp32 master / m / v equal the serial trainer's BIT FOR BIT;
    generate() and a checkpoint right behind a step see the updated weights."""
    from unimp_amd.train import Trainer, save_checkpoint
    import tempfile, os
    cfg = P.TINY
    om, layout = P.build_oracle(cfg)
    batches = [{k: v.cuda() for k, v in P.make_batch(cfg, layout, seed=860 + i).items()} for i in range(5)]
    res = {}
    for mode in (False, True):
        hm = P.build_hip(cfg, om, layout)
        tr = Trainer(hm, layout.special(), lr=1e-3, lr_scheduler="constant", overlap_optimizer=mode)
        losses = [tr.step(b)[0].item() for b in batches]
        if mode:
            assert tr._opt_event is not None and hm._params_ready is not None          # an update is (possibly) still in flight
        with tempfile.TemporaryDirectory() as d:
            save_checkpoint(os.path.join(d, "w.pt"), hm, tr)                           # syncs by itself
            sd = torch.load(os.path.join(d, "w.pt"))
        tr.sync()
        res[mode] = (losses, tr.opt.flat_p.clone(), tr.opt.master.clone(), tr.opt.m.clone(), tr.opt.v.clone(), sd)
        tr.dp.remove()
    (l0, p0, ma0, m0, v0, sd0), (l1, p1, ma1, m1, v1, sd1) = res[False], res[True]
    assert l0 == l1, (l0, l1)
    assert torch.equal(p0, p1) and torch.equal(ma0, ma1) and torch.equal(m0, m1) and torch.equal(v0, v1)
    assert sd0.keys() == sd1.keys() and all(torch.equal(sd0[k], sd1[k]) for k in sd0)
    with pytest.raises(ValueError):
        Trainer(P.build_hip(cfg, om, layout), layout.special(), overlap_optimizer=True, graph=True, dense_head_backward=True)


@pytest.mark.parametrize("mean_over", ["ga", "stashed"])
def test_flush_steps_an_incomplete_accumulation_group(P, mean_over):
    """The loader ends inside an accumulation group (3 micro-batches, GA = 2): ``Trainer.flush()`` turns the buffered micro-batch into
    an optimizer step -- accelerate steps at the end of the dataloader with every loss already divided by the full GA
    (``mean_over="ga"``, the default: the partial group's gradient is sum / GA); ``"stashed"`` divides by the count seen.  Fused and
    sequential accumulation must agree with each other, and nothing may be left for the next epoch."""
    from unimp_amd.train import Trainer
    cfg = P.TINY
    om, layout = P.build_oracle(cfg)
    mbs = [{k: v.cuda() for k, v in P.make_batch(cfg, layout, seed=830 + i).items()} for i in range(3)]
    res = {}
    for mode in ("sequential", "fused"):
        hm = P.build_hip(cfg, om, layout)
        tr = Trainer(hm, layout.special(), lr=1e-3, lr_scheduler="constant", grad_accum=2, fuse_accum=mode == "fused")
        grads, real = [], tr.opt.step
        tr.opt.step = lambda lr=None, grad_scale=1.0: (grads.append(tr.opt.flat_g.float() * grad_scale), real(lr=lr, grad_scale=grad_scale))[1]
        for b in mbs:
            tr.step(b)
        assert len(grads) == 1
        out = tr.flush(mean_over=mean_over)
        assert out is not None and not out.pending and len(grads) == 2 and tr.sched_step == 2
        assert tr.flush() is None and not tr._stash and tr._micro % 2 == 0
        res[mode] = (grads, tr.opt.master.clone())
        tr.dp.remove()
    (gs, ms), (gf, mf) = res["sequential"], res["fused"]
    assert P.rel_l2(gf[1], gs[1]) <= 2e-2 and P.rel_l2(mf, ms) <= 5e-3
    # the partial group's gradient: one micro-batch, scaled 1 / GA ("ga") or 1 / 1 ("stashed")
    hm = P.build_hip(cfg, om, layout)
    one = Trainer(hm, layout.special(), lr=1e-3, lr_scheduler="constant")
    g1 = []
    real1 = one.opt.step
    one.opt.step = lambda lr=None, grad_scale=1.0: (g1.append(one.opt.flat_g.float() * grad_scale), real1(lr=lr, grad_scale=grad_scale))[1]
    # same weights as after the first optimizer step of the runs above are not available here: compare the SCALE on fresh weights instead
    fr = {}
    for mo in ("ga", "stashed"):
        hm2 = P.build_hip(cfg, om, layout)
        t2 = Trainer(hm2, layout.special(), lr=1e-3, lr_scheduler="constant", grad_accum=2, fuse_accum=True)
        gg, r2 = [], t2.opt.step
        t2.opt.step = lambda lr=None, grad_scale=1.0, gg=gg, t2=t2, r2=r2: (gg.append(t2.opt.flat_g.float() * grad_scale), r2(lr=lr, grad_scale=grad_scale))[1]
        t2.step(mbs[2]); t2.flush(mean_over=mo)
        fr[mo] = gg[0]
        t2.dp.remove()
    one.step(mbs[2])
    assert P.rel_l2(fr["stashed"], g1[0]) <= 2e-2 and P.rel_l2(fr["ga"] * 2.0, g1[0]) <= 2e-2
    one.dp.remove()


@pytest.mark.parametrize("cfgname,round_to,rope_epilogue", [("TINY", 8, True), ("TINY_PAR", 8, True), ("TINY_MPT", 8, True), ("TINY_OPT", 8, True), ("CFG2_SLIM", 64, False), ("CFG2_SLIM", 64, True)])
def test_packed_token_order_equals_padded(P, monkeypatch, cfgname, round_to, rope_epilogue):
    """Trainer(packed=True): the language tower runs on the valid tokens only -- LayerNorm, the QKV / out / MLP / gated feed-forward
    projections on the packed rows, the attention kernels on the sequences as row ranges of the packed buffers (q_row_off / k_row_off),
    the rotation with each row's position from a table (the QKV GEMM's rotary epilogue where the padded path takes it too -- cfg2
    width, rope_epilogue True -- else the table pass).  Against the padded run on the same weights and batch: the same labels, logits
    equal at every VALID position BITWISE (every valid row goes through the same arithmetic), zero hidden state behind the <PAD>
    logits, the same loss, every gradient equal up to the summation order of the weight-gradient GEMMs (their contraction runs over
    another row order)."""
    from unimp_amd import functional as F_
    from unimp_amd.train import Trainer
    cfg = getattr(P, cfgname)
    om, layout = P.build_oracle(cfg)
    batch = {k: v.cuda() for k, v in P.make_batch(cfg, layout, seed=31).items()}
    monkeypatch.setattr(F_, "PACK_ROUND", round_to)
    monkeypatch.setattr(F_, "ROPE_EPILOGUE", rope_epilogue)   # off: both runs rotate q / k from the fp32 tables; on: both in the QKV GEMM's epilogue where it applies
    assert F_.Pack(batch["attention_mask"]).useful, "rounding leaves no row to skip: nothing is tested"
    res = {}
    for packed in (False, True):
        hm = P.build_hip(cfg, om, layout)
        tr = Trainer(hm, layout.special(), lr=1e-3, gamma=2.0, packed=packed)
        hm.train()
        loss, stats, out, labels = tr.forward_loss(batch)
        tr._backward(loss)
        res[packed] = (out["logits"].float(), loss.item(), labels.clone(), {n: p.grad.float().clone() for n, p in hm.named_parameters() if p.grad is not None})
        tr.dp.remove()
    monkeypatch.setattr(F_, "PACKED", False)
    valid = batch["attention_mask"].bool()
    assert int(valid.sum()) < valid.numel(), "the batch has no padding: nothing is tested"
    (la, lossa, laba, ga), (lb, lossb, labb, gb) = res[False], res[True]
    assert torch.equal(laba, labb)
    same_rope = True
    if same_rope:
        assert torch.equal(la[valid], lb[valid]), float((la[valid] - lb[valid]).abs().max())
        assert lossa == lossb
    else:
        assert P.rel_l2(lb[valid], la[valid]) <= 4e-3, P.rel_l2(lb[valid], la[valid])
        assert abs(lossa - lossb) <= 2e-4 * abs(lossa)
    # <PAD> rows of the packed run: final LayerNorm of a zero row -> beta -> the same logits at every <PAD> position
    pad_logits = lb[~valid]
    assert torch.equal(pad_logits, pad_logits[:1].expand_as(pad_logits))
    worst = (-1.0, "")
    for n, g in ga.items():
        assert n in gb, n
        if g.abs().max() == 0:
            assert gb[n].abs().max() == 0, n
            continue
        e = P.rel_l2(gb[n], g)
        worst = max(worst, (e, n))
        assert e <= (1e-2 if same_rope else 3e-2), (n, e)
    print(f"\\n[packed {cfgname}] valid rows {int(valid.sum())} of {valid.numel()}; loss {lossb:.6f} vs padded {lossa:.6f}; worst gradient rel-L2 {worst[0]:.2e} ({worst[1]})")
