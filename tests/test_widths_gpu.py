"""BASELINE's headline configurations at their REAL WIDTH against the fp32 oracle (reduced depth so the oracle runs in
tens of seconds): cfg2 (4b-instruct, single-task rec), cfg3 (the same model on a task-mixed batch: per-sample loss weights
2.0 / 1.0, rec_dataset.py:452) and cfg5's model family on its own workload (MPT-7B widths, head dim 128, ALiBi, an
image-token generation sequence of ~860 tokens padded to L = 1024: rec_dataset.py:613-664).

Every kernel shape of the bench meets the oracle here end to end: H = 2560 / hd = 80 / FFN 10 240 / V = 74 053 (odd) /
L = 512 / T = 8 x 257 ViT tokens / 512-key segment-masked cross attention.

Tolerances (north_star: loss 1e-3 rel, argmax bit-exact; the HIP path stores bf16 activations between kernels):
  labels   bit-exact
  loss     relative error <= 1e-3
  logits   rel-L2 <= max(1e-2, 1.5 x the deviation of the SAME oracle re-run under bf16 autocast)
  argmax   agreement rate over ALL valid positions is printed; identical wherever the HIP path's own top-2 margin
           exceeds 8 sigma of its measured logit error
  grads    per trainable tensor rel-L2 <= max(3e-2, 5 x that tensor's bf16-autocast noise floor); a tensor that misses it gets a
           second noise sample (the oracle at the product's storage precision); the scalar tanh gates -- one heavily cancelling
           dot product each -- additionally 0.2
Round 3 adds the STORAGE-PRECISION MODEL: the fp32 oracle with every point where the product writes a tensor to HBM rounded to
bf16 (oracle/numerics.py; the per-class budget is profiles/r03_error_budget_cfg2_slim.txt).  Its own deviation from pure fp32
(7.9e-3 at cfg2 width) is what ANY pipeline with these storage points shows; the two realisations of the rounding noise
decorrelate (a one-ulp difference upstream flips later roundings), so the product is not closer to the model than to fp32 -- but
its error MAGNITUDE must be the model's:
  logits   rel-L2 vs pure fp32 <= STORAGE_MODEL_SLACK x (the model's rel-L2 vs pure fp32): the kernels add nothing measurable on
           top of the storage rounding (measured ratios 0.98-1.00); the absolute 1e-2 bound stays as a second line
and ``test_argmax_exact_where_the_model_is_confident``: with a head in which every position has a clear winner (planted
rank-one terms, as a trained model has) the argmax must agree on 100 % of the valid positions.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
bf16 = torch.bfloat16


STORAGE_MODEL_SLACK = 1.10       # product's logits error vs fp32 <= 1.10 x the storage-precision model's own (measured 0.98-1.00)


@pytest.fixture(scope="module")
def P():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import _parity
    return _parity


@pytest.fixture(scope="module")
def slim2(P):
    om, layout = P.build_oracle(P.CFG2_SLIM)
    hm = P.build_hip(P.CFG2_SLIM, om, layout)
    return om, hm, layout


def _check_step(P, om, hm, layout, batch, name, gamma=2.0, reweight=True):
    from unimp_amd.train import Trainer
    want_logits, want_loss, want_labels, want_grads = P.oracle_step(om, layout, batch, gamma, reweight)
    tr = Trainer(hm, layout.special(), lr=1e-4, gamma=gamma, use_reweight=reweight)
    try:
        dev = {k: v.cuda() for k, v in batch.items()}
        hm.train()
        loss, stats, out, labels = tr.forward_loss(dev)
        assert torch.equal(labels.cpu(), want_labels), "label mask"
        got = out["logits"].float().cpu()
        assert got.shape == want_logits.shape
        floor = P.bf16_logit_floor(om, batch)
        e = P.rel_l2(got, want_logits)
        lerr = abs(loss.item() - want_loss.item()) / abs(want_loss.item())
        valid = batch["attention_mask"].bool()
        ag = P.argmax_agreement(got, want_logits, valid)
        print(f"\n[{name}] logits rel-L2 {e:.3e} (oracle bf16-autocast floor {floor:.3e}); loss {loss.item():.5f} vs {want_loss.item():.5f} "
              f"(rel {lerr:.2e}); argmax agreement {ag['rate']:.4f} over {ag['n']} valid positions, "
              f"{ag['n_sure']} with margin > 8 sigma ({ag['sigma']:.3e}): identical = {ag['sure_equal']}")
        assert e <= max(1e-2, 1.5 * floor), f"logits rel L2 {e} (floor {floor})"
        assert lerr <= 1e-3, (loss.item(), want_loss.item())
        from oracle import numerics as N_
        with torch.no_grad(), N_.storage(*N_.ALL):
            same = om(batch["vision_x"], batch["lang_x"], batch["attention_mask"])["logits"]
        e_same, e_model = P.rel_l2(got, same), P.rel_l2(same, want_logits)
        ag2 = P.argmax_agreement(got, same, valid)
        print(f"[{name}] storage-precision model (fp32 oracle, bf16 at the product's storage points): model vs fp32 {e_model:.3e} -> product / "
              f"model error ratio {e / e_model:.3f}; product vs model {e_same:.3e}, argmax agreement with the model {ag2['rate']:.4f}")
        assert e <= STORAGE_MODEL_SLACK * e_model, (e, e_model)
        assert e_same <= 1.5 * e_model, (e_same, e_model)       # two decorrelated realisations of the same noise: ~sqrt(2) x at most
        assert ag["n_sure"] > 0 and ag["sure_equal"], ag
        assert ag["rate"] >= 0.9, ag
        loss.backward()
        noise = P.bf16_noise_floor(om, layout, batch, want_labels, want_grads, gamma, reweight)
        second = {}

        def second_sample(n_):
            """a second, independent sample of the same noise, taken only when a tensor misses its first bound: the oracle's gradient
            at the product's storage precision.  The scalar tanh gates are ONE heavily cancelling dot product each -- a single noise
            sample is not a bound for them (cfg4's ff_gate: autocast sample 2.2e-2, the product 1.5e-1)."""
            if not second:
                with N_.storage(*N_.ALL):
                    _, _, _, mg = P.oracle_step(om, layout, batch, gamma, reweight)
                second.update({k: P.rel_l2(v, want_grads[k]) for k, v in mg.items() if k in want_grads and want_grads[k].norm() > 0})
            return second.get(n_, 0.0)
        named = dict(hm.named_parameters())
        worst, checked = (0.0, None), 0
        for n, g in want_grads.items():
            p = named[n]
            assert p.grad is not None, n
            if g.abs().max() == 0:
                assert p.grad.float().abs().max() == 0, n
                continue
            ge = P.rel_l2(p.grad, g)
            tol = max(3e-2, 5 * noise[n])
            if ge > tol:
                noise[n] = max(noise[n], second_sample(n))
                tol = max(3e-2, 5 * noise[n], 0.2 if g.numel() == 1 else 0.0)
            if ge / tol > worst[0]:
                worst = (ge / tol, f"{n}: {ge:.3e} (floor {noise[n]:.3e})")
            assert ge <= tol, f"grad {n}: rel L2 {ge} (bf16 noise floor {noise[n]})"
            checked += 1
        print(f"[{name}] {checked} gradient tensors within max(3e-2, 5 x floor); closest to its bound: {worst[1]}")
        assert checked >= 25
    finally:
        tr.dp.remove()
        hm.zero_grad(set_to_none=True)


def test_cfg2_full_width_reduced_depth_vs_oracle(P, slim2):
    """mmrec.py:177-213 at cfg2's widths: single-task rec batch (weights 2.0), gamma-2 focal reweighting."""
    om, hm, layout = slim2
    batch = P.make_batch(P.CFG2_SLIM, layout)
    assert float(batch["weights"][0]) == 2.0
    _check_step(P, om, hm, layout, batch, "cfg2 slim")


def test_argmax_exact_where_the_model_is_confident(P, slim2):
    """north_star: "token-id argmax bit-exact".  A random-init head over 74 053 tokens has near-ties at a few per cent of the
    positions whatever the arithmetic (that is what the 96 % of the test above measures).  A trained model is confident: here
    every valid position (with a hidden state of its own) gets a planted winner -- the head rows t_i gain dW with
    dW_i . h_j = alpha * delta_ij for the oracle's final hidden states h (least squares; distinct tokens, bf16-representable
    weights, the SAME head in both models) -- so its top-2 margin is many sigma of the logit error, and the HIP argmax must equal
    the oracle's at 100 % of those positions."""
    om, hm, layout = slim2
    cfg = P.CFG2_SLIM
    batch = P.make_batch(cfg, layout, seed=4321)
    grab = {}
    hk = om.lang_encoder.gpt_neox.final_layer_norm.register_forward_hook(lambda m, i, o: grab.__setitem__("h", o.detach()))
    head_o = om.lang_encoder.get_output_embeddings().weight
    head_h = hm.lang_encoder.get_output_embeddings().weight
    keep_o, keep_h = head_o.data.clone(), head_h.data.clone()
    try:
        with torch.no_grad():
            base = om(batch["vision_x"], batch["lang_x"], batch["attention_mask"])["logits"]
            hk.remove()
            hfin = grab["h"].reshape(-1, grab["h"].shape[-1])                     # [B*L, H]
            valid = batch["attention_mask"].bool().reshape(-1)
            pos = valid.nonzero()[:, 0]
            hv = hfin[pos].double()
            # positions whose hidden state repeats an earlier one (the BOS row of every sample ...) cannot have their own winner
            d2 = torch.cdist(hv, hv)
            dup = ((d2 < 1e-3 * hv.norm(dim=1, keepdim=True)) & torch.ones_like(d2, dtype=torch.bool).tril(-1)).any(1)
            pos, hv = pos[~dup], hv[~dup]
            g = torch.Generator().manual_seed(0)
            tok = torch.randperm(layout.base_vocab - 1, generator=g)[:pos.numel()] + 1         # distinct winners
            alpha = 8.0 * float(base.std())        # the largest of a row's 74 053 random logits sits ~4.3 sigma up: the winner clears it by a few sigma
            # rows dW with dW_i . h_j = alpha * delta_ij (least squares: n positions < H dims)
            head_o.data[tok] += (alpha * torch.linalg.pinv(hv).T).float()
            head_o.data.copy_(head_o.data.to(bf16).float())
            head_h.data.copy_(head_o.data.to(bf16))
            want = om(batch["vision_x"], batch["lang_x"], batch["attention_mask"])["logits"]
            hm.eval()
            got = hm(batch["vision_x"].cuda(), batch["lang_x"].cuda(), batch["attention_mask"].cuda())["logits"].float().cpu()
        wv, gv = want.reshape(-1, want.shape[-1])[pos], got.reshape(-1, got.shape[-1])[pos]
        assert (wv.argmax(-1) == tok).float().mean() > 0.9                         # the plant took
        top2 = wv.topk(2, -1).values
        margin = top2[:, 0] - top2[:, 1]
        sigma = float((gv - wv).std())
        agree = (gv.argmax(-1) == wv.argmax(-1))
        print(f"\n[confident head] {pos.numel()} valid positions; top-2 margin min {float(margin.min()):.3f} / median {float(margin.median()):.3f}; "
              f"logit error sigma {sigma:.3e} (min margin = {float(margin.min()) / sigma:.1f} sigma); argmax identical at {int(agree.sum())} / {agree.numel()}")
        assert float(margin.min()) > 8 * sigma, "the planted head is not confident enough for the claim"
        assert bool(agree.all()), f"argmax differs at {int((~agree).sum())} of {agree.numel()} confident positions"
    finally:
        hk.remove()
        head_o.data.copy_(keep_o)
        head_h.data.copy_(keep_h)


def test_cfg3_task_mixed_batch_vs_oracle(P, slim2):
    """cfg3 (unimp_all_tasks.sh): samples of different tasks share a batch, loss weight 2.0 for rec and 1.0 for the other
    tasks (rec_dataset.py:452); plus the unweighted, non-focal loss (mmrec.py:203 without --use_reweight)."""
    om, hm, layout = slim2
    batch = P.make_batch(P.CFG2_SLIM, layout, seed=77)
    batch["weights"] = torch.tensor([2.0, 1.0])
    _check_step(P, om, hm, layout, batch, "cfg3 mixed weights")
    _check_step(P, om, hm, layout, batch, "cfg3 mixed weights, no reweight", gamma=0.0, reweight=False)


def test_cfg4_hm_shapes_vs_oracle(P):
    """cfg4 (BASELINE: "H&M dataset path, 16 history images per user, gamma-focal loss"; unimp_hm.sh): T = 16 images -> 1024
    segment-masked media keys per text row, V = 66 216 (14 901 items), one optimizer-step's forward + loss + backward against the
    fp32 oracle at full width (round 2 only property-checked this configuration)."""
    cfg = P.CFG4_SLIM
    om, layout = P.build_oracle(cfg)
    assert layout.vocab == 66216
    hm = P.build_hip(cfg, om, layout)
    batch = P.make_batch(cfg, layout, seed=404)
    assert batch["vision_x"].shape[1] == 16 and int((batch["lang_x"] == layout.media).sum(1).min()) == 16
    _check_step(P, om, hm, layout, batch, "cfg4 slim (T = 16, V = 66 216)", gamma=2.0, reweight=True)


def _imggen_batch(layout, L, T, seed=5):
    """image-token generation sample (rec_dataset.py:613-664): unimp_amd.synthetic.make_imggen_batch with bf16-representable pixels."""
    from unimp_amd.synthetic import make_imggen_batch
    b = make_imggen_batch(layout, 1, T, L, seed=seed)
    b["vision_x"] = b["vision_x"].to(bf16).float()
    return b


def test_cfg5_mpt_width_image_generation_batch_vs_oracle(P):
    """cfg5's tower family on cfg5's workload in bf16: MPT-7B widths (32 heads of 128, ALiBi, tied head), causal attention
    over a 1024-token image-generation sequence with 257 labeled positions (eval_img_gen.py:102-111 generates the same span)."""
    cfg = P.CFG5_SLIM
    om, layout = P.build_oracle(cfg)
    hm = P.build_hip(cfg, om, layout)
    batch = _imggen_batch(layout, cfg["L"], cfg["T"])
    n_lab = 257
    _check_step(P, om, hm, layout, batch, "cfg5 slim (MPT widths, img-gen batch)")
    from oracle import train_step as ots
    sp = layout.special()
    labels = ots.label_mask_loop(batch["lang_x"].numpy(), sp["answer_id"], sp["eoc_id"], sp["pad_id"], sp["media_id"])
    assert int((labels != -100).sum()) == n_lab


def test_cfg5_fp8_frozen_towers_vs_error_model(P, monkeypatch):
    """cfg5 ("fp8 MFMA weights") at MPT-7B WIDTH (K = 4096 / 16384 contractions; round 2 checked tiny dims only, with bounds of 0.25):
    the frozen Linear layers on the MX-fp8 GEMMs (e4m3 + E8M0 per 32 k, weights quantised once, activations and dy on the fly).
    e4m3 carries 3 mantissa bits, so against fp32 this path is an order of magnitude coarser than bf16 BY CONSTRUCTION; what a kernel
    can be held to is the format's own error model: the fp32 oracle with the same products quantised the same way
    (oracle/numerics.py: mx_frozen) and the bf16 storage points on top.  Measured values are printed; asserted:
      * HIP fp8 vs the emulated-fp8 oracle: logits rel-L2 <= FP8_VS_MODEL x (HIP vs fp32) -- the two share the deterministic part of
        the error (the once-quantised weights), not the realisation of the activation roundings (measured 0.48);
      * HIP fp8 vs fp32 oracle no worse than 1.25 x what the error model itself deviates from fp32 (logits and every gradient);
      * loss within 2e-2 of the fp32 oracle."""
    from unimp_amd import functional as F_
    from unimp_amd.train import Trainer
    from oracle import numerics as N_
    cfg = P.CFG5_SLIM
    om, layout = P.build_oracle(cfg)
    batch = _imggen_batch(layout, cfg["L"], cfg["T"])
    want_logits, want_loss, want_labels, want_grads = P.oracle_step(om, layout, batch)
    with N_.mx_frozen(), N_.storage(*N_.ALL):
        mod_logits, mod_loss, _, mod_grads = P.oracle_step(om, layout, batch)
    monkeypatch.setattr(F_, "FP8_FROZEN", True)
    hm = P.build_hip(cfg, om, layout)
    tr = Trainer(hm, layout.special(), lr=1e-4, gamma=2.0)
    try:
        hm.train()
        loss, stats, out, labels = tr.forward_loss({k: v.cuda() for k, v in batch.items()})
        got = out["logits"].float().cpu()
        loss.backward()
        e_fp32, e_model, m_fp32 = P.rel_l2(got, want_logits), P.rel_l2(got, mod_logits), P.rel_l2(mod_logits, want_logits)
        l_fp32 = abs(loss.item() - want_loss.item()) / abs(want_loss.item())
        valid = batch["attention_mask"].bool()
        ag = P.argmax_agreement(got, mod_logits, valid)
        print(f"\\n[cfg5 fp8] logits rel-L2: HIP vs fp32 oracle {e_fp32:.3e}; HIP vs emulated-fp8 oracle {e_model:.3e}; emulated vs fp32 {m_fp32:.3e}; "
              f"loss {loss.item():.5f} vs fp32 {want_loss.item():.5f} ({l_fp32:.2e}) vs model {mod_loss.item():.5f}; argmax vs model {ag['rate']:.4f}")
        assert e_model <= FP8_VS_MODEL * e_fp32, (e_model, e_fp32)
        assert e_fp32 <= 1.25 * m_fp32 + 1e-3, (e_fp32, m_fp32)
        assert l_fp32 <= 2e-2
        named = dict(hm.named_parameters())
        worst = (0.0, None)
        for n, g in want_grads.items():
            if g.abs().max() == 0 or g.numel() == 1:
                continue
            ge, gm, gx = P.rel_l2(named[n].grad, g), P.rel_l2(mod_grads[n], g), P.rel_l2(named[n].grad, mod_grads[n])
            if ge / max(gm, 1e-3) > worst[0]:
                worst = (ge / max(gm, 1e-3), f"{n}: HIP vs fp32 {ge:.3e}, model vs fp32 {gm:.3e}, HIP vs model {gx:.3e}")
            assert ge <= 1.25 * gm + 3e-2, (n, ge, gm)
        print(f"[cfg5 fp8] gradient tensor closest to its bound (1.25 x the error model's own deviation + 3e-2): {worst[1]}")
    finally:
        tr.dp.remove()


FP8_VS_MODEL = 0.75              # HIP-vs-model error as a fraction of HIP-vs-fp32 (measured 0.48 on MI355X: DESIGN.md section 5.4)



def test_cfg5_width_fp8_loss_curve_against_the_chaos_floor(P, monkeypatch):
    """the fp8 loss-curve test at cfg5's WIDTH (MPT-7B dims, 2 of 32 blocks, one gated block, L = 1024) instead of
    the toy tower, with bounds read off a measured noise floor instead of a flat 25 %.  Three 16-step runs from identical weights, a FRESH
    batch every step (at this width the model memorises a b = 1 batch in one visit: a cycled pool's loss is 0.000 from the second pass on,
    tools/fp8curve_cfg5.py): bf16 (A), bf16 with ONE trainable weight moved by one bf16 ulp (B: how far two bf16 runs part by
    themselves -- the chaos floor, measured 5e-4), fp8 frozen towers (C).  The fp8 effect is NOT chaos: it perturbs every step's forward
    by the e4m3 error (logits 1.1e-1 rel-L2 at this width, test_cfg5_fp8_vs_error_model), which moves a 74 k-way focal loss by 0.1-1.8 %
    per step (measured).  Asserted: step 0 (same weights) within 0.5 %; every step within 4 %; mean gap within 2 %; and the floor itself
    below 0.5 % (if two bf16 runs part further, the comparison means nothing)."""
    from unimp_amd import functional as F_
    from unimp_amd.train import Trainer
    cfg = P.CFG5_SLIM
    om, layout = P.build_oracle(cfg)
    batches = [{k: v.cuda() for k, v in P.make_batch(cfg, layout, seed=950 + i).items()} for i in range(16)]
    curves = {}
    for name, fp8, nudge in (("A", False, False), ("B", False, True), ("C", True, False)):
        monkeypatch.setattr(F_, "FP8_FROZEN", fp8)
        hm = P.build_hip(cfg, om, layout)
        if nudge:
            w = next(p for n, p in hm.named_parameters() if p.requires_grad and p.dim() == 2)
            with torch.no_grad():
                v = w.view(-1)[:1].view(torch.int16)
                v += 1                                               # one ulp of one bf16 weight
        tr = Trainer(hm, layout.special(), lr=5e-5, lr_scheduler="constant", gamma=2.0)
        curves[name] = [tr.step(b)[0].item() for b in batches]
        tr.dp.remove()
        del tr, hm
        torch.cuda.empty_cache()
    monkeypatch.setattr(F_, "FP8_FROZEN", False)
    a, b, c = curves["A"], curves["B"], curves["C"]
    rel = lambda x, y: [abs(p_ - q_) / abs(p_) for p_, q_ in zip(x, y)]
    floor, gaps = rel(a, b), rel(a, c)
    print(f"\n[cfg5 width, fp8 loss curve] bf16 {' '.join(f'{x:.2f}' for x in a)}\n                              fp8  {' '.join(f'{x:.2f}' for x in c)}\n"
          f"  step-0 gap {gaps[0]:.2e}; largest gap fp8 vs bf16 {max(gaps):.2e}, mean {sum(gaps) / len(gaps):.2e}; chaos floor (one-ulp nudge) largest {max(floor):.2e}")
    assert min(a) > 1.0, "a loss near zero makes relative gaps meaningless: the batches must stay fresh"
    assert max(floor) <= 5e-3, floor
    assert gaps[0] <= 5e-3, gaps[0]
    assert max(gaps) <= 4e-2 and sum(gaps) / len(gaps) <= 2e-2, gaps
