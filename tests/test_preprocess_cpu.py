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
