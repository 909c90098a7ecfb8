"""Where does a sample's forward first depend on its batch?  Builds the 9b model (the configuration whose bitwise b = 1 vs b = 2 check in
tests/test_fullsize_gpu.py::test_cfg5_9b_mpt_tower_train_step failed once in four runs), hooks the vision encoder, the Perceiver and
every decoder layer, and compares sample 1 inside a batch of 2 with sample 1 alone, several trials, plus the same batch twice
(run-to-run determinism).  Prints the first module whose output differs.  usage: debug_batch_invariance.py [4b|9b] [trials]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                  # noqa: E402
from unimp_amd.synthetic import make_batch    # noqa: E402
from unimp_amd import ops                     # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "9b"
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda")
model, layout = bench.build_cfg2(dev, gate=0.5, lang="anas-awadalla/mpt-7b", every=4) if which == "9b" else bench.build_cfg2(dev, gate=0.5)
model.eval()
names, store = [], {}


def hook(name):
    def f(m, i, o):
        t = o[0] if isinstance(o, (tuple, list)) else o
        if name == "vision_encoder" and isinstance(o, (tuple, list)):
            t = o[1]                               # (pooled, tokens): the tokens feed the Perceiver
        if torch.is_tensor(t):
            store[name] = t.detach().clone()
    return f


model.vision_encoder.register_forward_hook(hook("vision_encoder"))
model.perceiver.register_forward_hook(hook("perceiver"))
for i, layer in enumerate(model.lang_encoder._get_decoder_layers()):
    layer.register_forward_hook(hook(f"layer{i:02d}"))
    if getattr(layer, "gated_cross_attn_layer", None) is not None:
        layer.gated_cross_attn_layer.register_forward_hook(hook(f"layer{i:02d}.xattn"))


def run(bt):
    store.clear()
    with torch.no_grad():
        lg = model(bt["vision_x"], bt["lang_x"], bt["attention_mask"])["logits"]
    out = dict(store)
    out["logits"] = lg.clone()
    return out


def sample1(t, B, like):
    if t.shape[0] == like.shape[0]:            # not batch-shaped
        return t
    if t.shape[0] == B * like.shape[0]:        # [B, ...] or [B*T, ...] rows: the last sample's share
        return t[(B - 1) * like.shape[0]:]
    raise ValueError((t.shape, like.shape))


for trial in range(trials):
    bt = make_batch(layout, 2, 8, 512, seed=5 + trial, device="cuda", vision_dtype=torch.bfloat16)
    a = run(bt)
    a2 = run(bt)
    one = {k: v[1:] for k, v in bt.items()}
    c = run(one)
    keys = [k for k in a if k in c]
    first_rr = next((k for k in sorted(keys, key=lambda s: (s == "logits", s)) if not torch.equal(a[k], a2[k])), None)
    first_bi = next((k for k in sorted(keys, key=lambda s: (s == "logits", s)) if not torch.equal(sample1(a[k], 2, c[k]), c[k])), None)
    worst = max((float((sample1(a[k], 2, c[k]).float() - c[k].float()).abs().max()), k) for k in keys)
    print(f"trial {trial}: run-to-run first difference: {first_rr}; batch-of-2 vs alone first difference: {first_bi}; largest |diff| {worst[0]:.4g} at {worst[1]}; "
          f"live-tuned shapes so far {len(ops.TUNE_MISSES)}", flush=True)
