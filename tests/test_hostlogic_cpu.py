"""Host-side decisions of the GEMM / RoPE routing (pure functions, no GPU): the plans the bench's shapes get, and the
pair-adjacent row permutation of the rotary epilogue against the half-split rotation it replaces (oracle/lm.py)."""
import math
import os
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tail_split_plan_for_the_step_shapes():
    from unimp_amd.ops import _tail_split_plan
    # the gated blocks' two big weight gradients at b = 64 (400 tiles): one full round + a split-K remainder in whole rounds
    for M, N in ((10240, 2560), (2560, 10240)):
        axis, cut, S = _tail_split_plan(M, N, 32768)
        assert axis == ("m" if M > N else "n") and cut % 256 == 0
        ta = (cut // 256) * ((min(M, N) + 255) // 256)
        tb = 400 - ta
        assert ta <= 256 and tb > 0 and (tb * S) % 256 >= 200 or (tb * S) % 256 == 0, (ta, tb, S)
    assert _tail_split_plan(16384, 4096, 24576) is None          # 1024 tiles: whole rounds already
    assert _tail_split_plan(4096, 4096, 24576) is None           # exactly one round
    assert _tail_split_plan(10240, 2560, 4096) is None           # shallow K: a slab round trip costs more than the idle half round


def test_splitk_count_balances_rounds_and_slab_traffic():
    from unimp_amd.ops import _splitk_count
    for M, N, K in ((2560, 512, 32768), (1024, 1024, 32768), (1024, 4096, 32768), (640, 2560, 74053), (512, 1024, 2120)):
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        S = _splitk_count(M, N, K, tiles)
        ks = (-(-K // S) + 63) // 64 * 64
        assert 2 <= S <= 32 and -(-K // ks) == S and ks >= 512, (M, N, K, S)
        assert tiles * S <= 2 * 256 + 64, (M, N, K, S)            # never more than ~two rounds of ever shorter blocks
    assert _splitk_count(2560, 512, 32768, 20) == 12             # one round of 240 blocks instead of two of 250 (measured 129 -> 109 us)


def test_rope_adjacent_permutation_is_the_half_split_rotation():
    """rotating adjacent pairs (8g + j, 8g + 4 + j) at frequency 4g + j of the permuted vector == permuting the half-split
    rotation of the original vector: the identity the rotary epilogue rests on (both QKV layouts, partial rotary)."""
    from unimp_amd.functional import _rope_perm_index
    from oracle.lm import neox_rope_tables, rotate_half
    for nh, hd, rot, inter in ((4, 80, 80, True), (3, 64, 16, True), (2, 128, 128, False)):
        H = nh * hd
        idx = _rope_perm_index(nh, hd, rot, inter, "cpu")
        assert sorted(idx.tolist()) == list(range(3 * H))
        v_rows = idx.view(nh, 3, hd)[:, 2] if inter else idx[2 * H:]
        want_v = (torch.arange(nh)[:, None] * 3 * hd + 2 * hd + torch.arange(hd)[None]) if inter else torch.arange(2 * H, 3 * H)
        assert torch.equal(v_rows.reshape(-1), want_v.reshape(-1))           # v keeps its order
        L = 7
        y = torch.randn(L, 3 * H)
        cos, sin = neox_rope_tables(L, rot, 10000.0)
        yv = y.view(L, nh, 3, hd) if inter else y.view(L, 3, nh, hd).permute(0, 2, 1, 3)
        want = yv.clone()
        for part in (0, 1):
            r = yv[:, :, part, :rot]
            want[:, :, part, :rot] = r * cos[:, None] + rotate_half(r) * sin[:, None]
        want = want.reshape(L, 3 * H) if inter else want.permute(0, 2, 1, 3).reshape(L, 3 * H)
        yp = y[:, idx]                                                       # what the permuted projection produces
        gv = yp.view(L, nh, 3, hd) if inter else yp.view(L, 3, nh, hd).permute(0, 2, 1, 3)
        got = gv.clone()
        theta = 10000.0 ** (-2.0 * torch.arange(rot // 2) / rot)
        ang = torch.arange(L)[:, None] * theta[None]                         # [L, rot / 2]
        for part in (0, 1):
            x = gv[:, :, part, :rot].reshape(L, nh, rot // 8, 2, 4)
            c, s_ = ang.cos().view(L, 1, rot // 8, 4), ang.sin().view(L, 1, rot // 8, 4)
            x1, x2 = x[..., 0, :], x[..., 1, :]
            got[:, :, part, :rot] = torch.stack([x1 * c - x2 * s_, x2 * c + x1 * s_], -2).reshape(L, nh, rot)
        got = got.reshape(L, 3 * H) if inter else got.permute(0, 2, 1, 3).reshape(L, 3 * H)
        assert torch.allclose(got, want[:, idx], atol=1e-5), (nh, hd, rot, inter)


def test_bench_launcher_command(monkeypatch):
    """bench.py --gpus N without a launcher environment: the parent builds a torch.distributed.run command on 127.0.0.1 for N
    ranks with its own arguments and returns the launcher's exit code; it never touches the GPU (none exists here)."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    try:
        bench.main()
        raise AssertionError("main() must exit with the launcher's code")
    except SystemExit as e:
        assert e.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_eval_text_metrics_hand_computed():
    """the host half of eval_exp / eval_img_sel (UniMP/pipeline/eval/eval_exp.py:116-142, eval_img_sel.py:96-113): parsing of the
    generated text and the metric definitions, on cases worked out by hand."""
    from unimp_amd import eval as E
    # eval_exp.py:116-125
    assert E.parse_rating_explanation("Describe? What is the rating? rate_4 solid and cheap") == (4.0, "solid and cheap")
    assert E.parse_rating_explanation("q? nonsense words") == (3.0, "words")           # first word is no rating: 3.0, still dropped
    assert E.parse_rating_explanation("q?") == (3.0, "Empty")
    assert E.parse_rating_explanation("q? rate_2") == (2.0, "Empty")
    # BLEU unigram modified precision, corpus level: clipped matches / predicted unigrams
    #   "the the the cat" vs "the cat sat": the clipped to 1, cat 1 -> 2 / 4;  "a dog." vs "a dog ." : 3 / 3 (13a splits the period)
    assert E.bleu1_precision(["the the the cat"], ["the cat sat"]) == 0.5
    assert abs(E.bleu1_precision(["the the the cat", "a dog."], ["the cat sat", "a dog ."]) - 5 / 7) < 1e-12
    assert E.bleu1_precision([""], ["x"]) == 0.0
    # ROUGE: pred "the cat sat on the mat" (6), ref "the cat is on the mat" (6): unigram overlap 5 -> P = R = 5/6; bigrams
    # (the cat) (on the) (the mat) of 5 -> 3/5; LCS = the cat on the mat = 5
    r = E.rouge_f(["the cat sat on the mat"], ["The cat is on the mat!"])
    assert abs(r["rouge1"] - 5 / 6) < 1e-12 and abs(r["rouge2"] - 0.6) < 1e-12 and abs(r["rougeL"] - 5 / 6) < 1e-12
    r = E.rouge_f(["a b c d"], ["d c b a"])                   # same words, reversed: unigrams 1.0, bigrams 0, LCS 1 -> 0.25
    assert r["rouge1"] == 1.0 and r["rouge2"] == 0.0 and r["rougeL"] == 0.25
    # METEOR (eval_exp.py:65,146), exact + Porter-stem stages, nltk's alpha 0.9 / beta 3 / gamma 0.5.  Porter's published examples first:
    for w, st in (("caresses", "caress"), ("ponies", "poni"), ("agreed", "agre"), ("plastered", "plaster"), ("motoring", "motor"), ("hopping", "hop"),
                  ("filing", "file"), ("happy", "happi"), ("sky", "sky"), ("relational", "relat"), ("rational", "ration"), ("digitizer", "digit"),
                  ("electrical", "electr"), ("replacement", "replac"), ("adoption", "adopt"), ("controll", "control"), ("roll", "roll"), ("sing", "sing")):
        assert E._porter_stem(w) == st, (w, E._porter_stem(w), st)
    # identical sentences: 6 matches in one chunk -> F = 1, penalty 0.5 (1/6)^3 -> 0.99769 (the value nltk documents for this case)
    assert abs(E.meteor(["the cat sat on the mat"], ["the cat sat on the mat"]) - (1 - 0.5 / 216)) < 1e-12
    # "the cats sat on the mat" vs "the cat is sitting on the mat": exact the / on / the / mat, stem cats ~ cat (sat / sitting do not share a
    # stem): m = 5, P = 5/6, R = 5/7, F = PR / (0.9 P + 0.1 R), two chunks -> penalty 0.5 (2/5)^3
    P_, R_ = 5 / 6, 5 / 7
    want = P_ * R_ / (0.9 * P_ + 0.1 * R_) * (1 - 0.5 * (2 / 5) ** 3)
    assert abs(E.meteor(["the cats sat on the mat"], ["the cat is sitting on the mat"]) - want) < 1e-12
    # the same six words in another order: every word matches, but the greedy alignment (walk the hypothesis from its end, take the LAST free
    # equal reference word) pairs the4 -> the4 and the1 -> the0: (0,3) (1,0) (2,5) (3,2) (4,4) (5,1), no two adjacent -> 6 chunks, penalty 0.5
    assert abs(E.meteor(["on the mat sat the cat"], ["the cat sat on the mat"]) - 0.5) < 1e-12
    assert E.meteor(["x y"], ["a b"]) == 0.0 and E.meteor([""], ["a"]) == 0.0
    # eval_img_sel.py:96-113: the SET of generated words
    assert E.selection_scores("Select? s_0 s_2 s_2", [0, 1]) == {"recall": 0.5, "precision": 0.5, "f1": 0.5}
    assert E.selection_scores("Select?", [1]) == {"recall": 0.0, "precision": 0, "f1": 0.0}
    assert E.selection_scores("x? s_1", [1]) == {"recall": 1.0, "precision": 1.0, "f1": 1.0}


def test_imggen_synthetic_batch_layout():
    """BASELINE config 5's workload generator (bench.py --task img_gen; rec_dataset.py:613-664): every sample has T <image> chunks, 855
    real tokens at T = 2, exactly the 256 code tokens + EOS of the target labeled, loss weight 1.0."""
    from unimp_amd.synthetic import TokenLayout, make_imggen_batch
    from oracle import train_step as ots
    lay = TokenLayout()
    b = make_imggen_batch(lay, 3, 2, 1024, image_size=8)
    sp = lay.special()
    lab = ots.label_mask(b["lang_x"].numpy(), sp["answer_id"], sp["eoc_id"], sp["pad_id"], sp["media_id"])
    assert b["vision_x"].shape == (3, 2, 1, 3, 8, 8) and b["weights"].tolist() == [1.0] * 3
    assert b["attention_mask"].sum(1).tolist() == [855] * 3 and ((lab != -100).sum(1) == 257).all()
    assert ((b["lang_x"] == lay.media).sum(1) == 2).all()
    img0 = lay.item0 + lay.n_items
    labeled = b["lang_x"].numpy()[lab != -100].reshape(3, 257)
    assert ((labeled[:, :256] >= img0) & (labeled[:, :256] < img0 + 1024)).all() and (labeled[:, 256] == lay.eos).all()


def test_pack_maps_right_padded_batches_to_row_ranges(monkeypatch):
    """functional.Pack (Trainer(packed=True)): the valid tokens of a right-padded batch (collate_rec.py:38-74, data.py:274) as row ranges of
    one buffer -- idx / inv are inverse maps, sequence b owns rows off[b] .. off[b] + len[b] - 1, pos is the position inside the
    sequence, M is the valid count rounded up; a mask that is not a prefix of ones per row is refused."""
    import pytest
    from unimp_amd import functional as F_
    lens = [7, 0, 12, 3]
    B, L = len(lens), 12
    mask = torch.zeros(B, L, dtype=torch.int64)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
    monkeypatch.setattr(F_, "PACK_ROUND", 8)
    pk = F_.Pack(mask)
    nv = sum(lens)
    assert pk.useful and pk.nv == nv and pk.M == 24 and pk.M % 8 == 0
    want = [b * L + j for b, n in enumerate(lens) for j in range(n)]
    assert pk.idx[:nv].tolist() == want
    assert all(mask.reshape(-1)[i] == 0 for i in pk.idx[nv:].tolist())          # the spare rows point at a <PAD> slot
    assert pk.inv[want].tolist() == list(range(nv)) and int((pk.inv >= 0).sum()) == nv
    assert pk.rows.len.tolist() == lens and pk.rows.off.tolist() == [0, 7, 7, 19] and pk.rows.n == nv and pk.rows.S == L
    assert pk.pos[:nv].tolist() == [j for n in lens for j in range(n)]
    seg = torch.arange(B * L, dtype=torch.int32).view(B, L)
    assert pk.seg(seg)[:nv].tolist() == want
    monkeypatch.setattr(F_, "PACK_ROUND", None)                                  # default granularity: 1/16 of B * L, at least 256 rows
    assert not F_.Pack(mask).useful                                              # 22 valid tokens round up to all 48 rows: nothing to skip
    left = mask.flip(1)
    monkeypatch.setattr(F_, "PACK_ROUND", 8)
    with pytest.raises(ValueError):
        F_.Pack(left)


def test_packed_flag_is_per_tower_and_qk_ln_towers_keep_the_padded_path(monkeypatch):
    """packed token order is a property of the tower (lm._TowerBase.packed), not a process global that a second Trainer
    flips for the first; a tower without a packed-row form (MosaicGPT with QK-LayerNorm, Llama) never packs -- under UNIMP_PACKED=1 it
    keeps the padded path instead of raising NotImplementedError in the middle of a forward."""
    from unimp_amd import functional as F_, lm
    mask = torch.zeros(4, 64, dtype=torch.int64)
    mask[:, :5] = 1
    monkeypatch.setattr(F_, "PACK_ROUND", 8)
    neox = lm.build_lm(lm._Cfg(model_type="gpt_neox", vocab_size=64, hidden_size=32, num_hidden_layers=1, num_attention_heads=4,
                               intermediate_size=64, rotary_pct=1.0, rotary_emb_base=10000.0, layer_norm_eps=1e-5, max_position_embeddings=64,
                               use_parallel_residual=False))
    other = lm.build_lm(lm._Cfg(model_type="gpt_neox", vocab_size=64, hidden_size=32, num_hidden_layers=1, num_attention_heads=4,
                                intermediate_size=64, rotary_pct=1.0, rotary_emb_base=10000.0, layer_norm_eps=1e-5, max_position_embeddings=64,
                                use_parallel_residual=False))
    neox.train(); other.train()
    monkeypatch.setattr(F_, "PACKED", False)
    assert neox._pack(mask, None) is None
    neox.packed = True                                        # what Trainer(packed=True) sets
    assert neox._pack(mask, None) is not None and other._pack(mask, None) is None
    assert neox._pack(mask, object()) is None                 # cached decode never packs
    neox.eval()
    assert neox._pack(mask, None) is None
    cfg = lm.MosaicGPTConfig()
    assert cfg.attn_qk_ln
    monkeypatch.setattr(F_, "PACKED", True)                   # UNIMP_PACKED=1
    assert lm.LlamaForCausalLM.supports_packed is False
    class _T(lm._TowerBase):
        supports_packed = False
        training = True
    assert _T._pack(_T.__new__(_T), mask, None) is None


def test_autotune_table_ships_inside_the_package():
    """the product's default table is a file of the package, identical to the measured copy under profiles/."""
    import json
    from unimp_amd import ops
    pkg = os.path.join(ROOT, "unimp_amd", "gemm_autotune_gfx950.json")
    assert os.path.samefile(ops._TUNE_DEFAULT, pkg)
    a, b = json.load(open(pkg)), json.load(open(os.path.join(ROOT, "profiles", "gemm_autotune_gfx950.json")))
    assert a == b and len(a) > 100


def test_avoid_persistent_is_reference_counted():
    """two data-parallel trainers closed in creation order must not clear the flag while the second one lives."""
    from unimp_amd import ops
    assert not ops.AVOID_PERSISTENT
    ops.avoid_persistent_acquire()
    ops.avoid_persistent_acquire()
    ops.avoid_persistent_release()            # the FIRST trainer closes first
    assert ops.AVOID_PERSISTENT
    ops.avoid_persistent_release()
    assert not ops.AVOID_PERSISTENT
    ops.avoid_persistent_release()            # an extra release is harmless
    assert not ops.AVOID_PERSISTENT and ops._avoid_persistent_holders == 0
