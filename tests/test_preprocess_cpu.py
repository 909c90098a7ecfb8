"""Input-pipeline rows on the CPU: the oracle's Pillow restatement against the committed Pillow
outputs and (when Pillow is importable) against Pillow live; the product's tap-table builder against the oracle's; collate
against the reference's own collate_fn output (tests/golden/collate_rec.npz, oracle/make_golden_preprocess.py)."""
import os
import numpy as np
import pytest
import torch

from oracle import preprocess as OP
from oracle.make_golden_preprocess import synth_image
from unimp_amd import data as D

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_oracle_resize_matches_pillow_golden():
    g = np.load(os.path.join(GOLD, "preprocess_pillow.npz"))
    for seed, (H, W), want in zip(g["seeds"], g["shapes"], g["resized"]):
        got = OP.resize_bicubic_u8(synth_image(int(seed), int(H), int(W)), 224, 224)
        assert np.array_equal(got, want), (int(seed), int(H), int(W))


def test_oracle_resize_matches_pillow_live():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    for H, W in [(640, 480), (3, 1000), (60, 61), (224, 10), (900, 224)]:
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        want = np.asarray(Image.fromarray(img, "RGB").resize((224, 224), Image.BICUBIC))
        assert np.array_equal(OP.resize_bicubic_u8(img, 224, 224), want), (H, W)


@pytest.mark.parametrize("n", [1, 2, 3, 17, 100, 223, 224, 225, 375, 500, 640, 1000, 1600])
def test_product_tap_tables_equal_oracle(n):
    b, k, ks = OP.resample_coeffs(n, 224)
    t = D.bicubic_taps(n, 224)
    assert t.shape == (224, 2 + ks) and np.array_equal(t[:, :2], b) and np.array_equal(t[:, 2:], k)
    assert (t[:, 2:].sum(1) - (1 << 22)).__abs__().max() <= ks          # taps sum to 1.0 up to per-tap rounding


def test_to_tensor_normalize_definition():
    u8 = np.arange(2 * 2 * 3, dtype=np.uint8).reshape(2, 2, 3) * 20
    x = OP.to_tensor_normalize(u8)
    t = torch.from_numpy(u8).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    t = t.sub(torch.tensor(OP.FLAMINGO_MEAN)[:, None, None]).div(torch.tensor(OP.FLAMINGO_STD)[:, None, None])
    assert x.dtype == np.float32 and np.array_equal(x, t.numpy())


def test_collate_matches_reference_golden():
    g = np.load(os.path.join(GOLD, "collate_rec.npz"))
    ids, off, samples = torch.from_numpy(g["ids"]), 0, []
    for i, n in enumerate(g["lens"]):
        samples.append({"net_input": {"input_ids": ids[off:off + n], "attention_masks": torch.ones(int(n), dtype=torch.long),
                                      "patch_images": torch.from_numpy(g["images"][i]), "weights": torch.tensor(float(g["weights_in"][i]))}})
        off += int(n)
    b = D.collate_fn(samples, pad_idx=1, eos_idx=2)["net_input"]
    assert np.array_equal(b["input_ids"].numpy(), g["out_ids"]) and b["input_ids"].dtype == torch.int64
    assert np.array_equal(b["attention_masks"].numpy(), g["out_masks"])
    assert np.array_equal(b["weights"].numpy(), g["out_weights"])
    assert np.array_equal(b["patch_images"].numpy(), g["out_images"])
    assert D.collate_fn([], 1, 2) == {}
    left = D.collate_tokens([torch.tensor([5, 6, 2]), torch.tensor([7, 2])], 1, eos_idx=2, left_pad=True, move_eos_to_beginning=True)
    assert left.tolist() == [[2, 5, 6], [1, 2, 7]]


def test_rec_prompt_template():
    s = D.rec_prompt([(11, "Name a Color b"), (12, "Name c")], 13)
    assert s == ("<image> Name a Color b <answer> item_11 <|endofchunk|> <image> Name c <answer> item_12 <|endofchunk|> "
                 "What is the next item recommended to the user? <answer> item_13")


def _materialise_rec_dataset(tmp_path):
    g = np.load(os.path.join(GOLD, "rec_dataset.npz"))
    for i, name in enumerate(g["file_names"]):
        p = tmp_path / str(name)
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(g[f"file_{i}"].tobytes())
    return g


def _tokenizer():
    from unimp_amd.factory import SyntheticTokenizer
    tok = SyntheticTokenizer()
    tok.add_special_tokens({"additional_special_tokens": ["<|endofchunk|>", "<image>", "<answer>"]})
    tok.add_special_tokens({"pad_token": "<PAD>"})
    return tok


def test_rec_dataset_matches_reference_golden(tmp_path):
    """unimp_amd.data.RecDataset against samples produced by the reference's own RecDataset class on the same files, the
    same tokenizer and the same numpy seed (oracle/make_golden_preprocess.py --rec-dataset)."""
    pytest.importorskip("PIL.Image")
    from unimp_amd.factory import ImageProcessor
    g = _materialise_rec_dataset(tmp_path)
    for split, task in (("train", "rec"), ("test", "rec"), ("train", "search"), ("test", "search"), ("train", "exp"), ("test", "exp"),
                        ("train", "img_sel"), ("test", "img_sel"), ("train", "img_gen"), ("test", "img_gen")):
        ds = D.RecDataset(str(tmp_path), "all", _tokenizer(), split=split, image_transform=ImageProcessor(224), task=task, n_items=14)
        assert len(ds) == 4
        np.random.seed(11)
        if task != "rec":
            split = task + "_" + split
        for idx in range(3):
            s = ds[idx]
            ni = s["net_input"]
            assert np.array_equal(ni["input_ids"].numpy(), g[f"{split}{idx}_ids"]), (split, idx)
            assert np.array_equal(ni["attention_masks"].numpy(), g[f"{split}{idx}_mask"])
            assert np.array_equal(ni["patch_images"][:, :, ::16, ::16].numpy(), g[f"{split}{idx}_img_sub"])
            sums = np.array([float(ni["patch_images"].double().sum()), float(ni["patch_images"].double().abs().sum())])
            assert np.array_equal(sums, g[f"{split}{idx}_img_sum"])
            if split.endswith("train"):
                assert float(ni["weights"]) == float(g[f"{split}{idx}_w"]) == (2.0 if task == "rec" else 1.0)
            else:
                no = s["net_output"]
                if task == "exp":
                    assert f"{no['output_ratings']}|{no['output_exps']}" == str(g[f"{split}{idx}_target"])
                elif task == "img_sel":
                    assert no["output_ids"].tolist() == g[f"{split}{idx}_target"].tolist()
                elif task == "img_gen":
                    assert f"{no['output_ids']}|{no['items']}" == str(g[f"{split}{idx}_target"])
                else:
                    assert no["output_ids"] == str(g[f"{split}{idx}_target"])
                assert ni["input_len"] == int(g[f"{split}{idx}_input_len"])


def test_rec_dataset_deferred_images_and_collate(tmp_path):
    pytest.importorskip("PIL.Image")
    _materialise_rec_dataset(tmp_path)
    tok = _tokenizer()
    ds = D.RecDataset(str(tmp_path), "all", tok, split="train", defer_images=True)
    np.random.seed(11)
    samples = [ds[i] for i in range(3)]
    assert all(len(s["net_input"]["patch_images"]) == 5 and s["net_input"]["patch_images"][0].dtype == np.uint8 for s in samples)
    b = ds.collate(samples)["net_input"]
    L = max(s["net_input"]["input_ids"].numel() for s in samples)
    assert b["input_ids"].shape == (3, L) and b["attention_masks"].shape == (3, L) and b["weights"].tolist() == [2.0] * 3
    assert (b["input_ids"][b["attention_masks"] == 0] == tok.pad_token_id).all()
    assert len(b["patch_images_raw"]) == 3 and "patch_images" not in b
    with pytest.raises(ValueError):
        D.RecDataset(str(tmp_path), "all", tok)


def test_rec_metrics_match_reference_golden():
    from unimp_amd import eval as E
    g = np.load(os.path.join(GOLD, "rec_metrics.npz"))
    for r, want in zip(g["r"], g["metrics"]):
        got = [E.hit_at_k(r, k) for k in (3, 5, 10)] + [E.ndcg_at_k(r, k, 1) for k in (3, 5, 10)] + [E.mrr_at_k(r, k) for k in (3, 5, 10)]
        assert np.array_equal(np.array(got), want), r
    assert E.ndcg_at_k([0, 0, 1, 0, 0, 0, 0, 0, 0, 0], 10, 1) == 0.5          # hit at rank 3
    r = E.relevance(["a b? item_7", "q? item_9</s> junk", "item_7"], "item_7")
    assert r.tolist() == [1, 0, 1, 0, 0, 0, 0, 0, 0, 0]
    m = E.user_metrics(r)
    assert m["hr@3"] == 1.0 and m["mrr@10"] == 1.0


def test_mixed_task_dataset_matches_reference_mixture(tmp_path):
    """which users of which task end up in the multi-task training set, in which order: against the reference's own
    RecDataset(task=[...], single_task=False) on the same files and numpy seed."""
    import json
    g = _materialise_rec_dataset(tmp_path)
    for name in ("train_users.json", "train_all_exp.json", "train_all_img_sel.json"):     # 12 users per task file, as the generator made them
        d = json.loads((tmp_path / name).read_text())
        (tmp_path / name).write_text(json.dumps({f"{k}_{c}": v for c in range(3) for k, v in d.items()}))
    np.random.seed(5)
    mixed = D.MixedRecDataset(str(tmp_path), "all", _tokenizer(), ["img_sel", "search", "rec", "exp"], defer_images=True, n_items=14)
    assert mixed.tasks == g["mix_tasks"].tolist() and len(mixed) == len(g["mix_tasks"])
    seqs = [mixed.parts[t].seqs[j] for t, j in mixed.index]
    assert [int(sq[0][0]) for sq in seqs] == g["mix_first_items"].tolist()
    assert [len(sq) for sq in seqs] == g["mix_seq_lens"].tolist()
    np.random.seed(1)
    s = mixed[0]
    assert float(s["net_input"]["weights"]) == 1.0 and float(mixed[7]["net_input"]["weights"]) == 2.0       # img_sel vs rec
