"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

TEST INFRASTRUCTURE ONLY.  Needs /root/reference (read-only) -- it never travels to the GPU
box; only the small .npz fixtures written here do.  Re-run:  python oracle/make_golden.py

Fixtures (inputs + expected outputs; no reference source text is stored):
  train_step_g{0,2}_rw{0,1}.npz  mmrec.train_one_epoch driven with stubbed third-party modules:
                                 ids/weights/logits -> labels, optimised loss, dloss/dlogits
  clip_tiny.npz                  UniMP/xformers_model/clip.py CLIPVisionModel (xformers.ops stubbed
                                 with SDPA): pixels + weights -> last_hidden_state
  llama_tiny.npz                 UniMP/xformers_model/llama.py LlamaForCausalLM: ids + weights ->
                                 logits, loss, grads
  neox_tiny.npz / opt_tiny.npz   installed transformers GPTNeoX / OPT from config (third-party
                                 towers of the reference): ids + mask + weights -> logits
"""
import os
import sys
import types
import contextlib
from unittest.mock import MagicMock

import numpy as np
import torch

REF = "/root/reference/UniMP"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _np(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


# ------------------------------------------------------------------ xformers stub
def install_xformers_stub():
    import torch.nn.functional as F
    xf, ops = types.ModuleType("xformers"), types.ModuleType("xformers.ops")

    class LowerTriangularMask:
        pass

    def memory_efficient_attention(q, k, v, attn_bias=None, p=0.0, scale=None):
        o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2),
                                           is_causal=isinstance(attn_bias, LowerTriangularMask), scale=scale)
        return o.transpose(1, 2)

    ops.memory_efficient_attention, ops.LowerTriangularMask = memory_efficient_attention, LowerTriangularMask
    xf.ops = ops
    sys.modules["xformers"], sys.modules["xformers.ops"] = xf, ops


# ------------------------------------------------------------------ 1. train_one_epoch capture
def gen_train_step():
    import transformers, datasets  # noqa: F401  (import real ones before mocking the rest)
    for m in ["wandb", "open_flamingo", "braceexpand", "torchvision", "torchvision.transforms",
              "torchvision.transforms.functional", "torchvision.datasets", "webdataset", "webdataset.filters",
              "webdataset.tariterators", "evaluate", "nltk", "prettytable", "pycocoevalcap",
              "pycocoevalcap.eval", "pycocotools", "pycocotools.coco", "open_clip", "utils", "metric",
              "metric.evaluator", "deepspeed"]:
        sys.modules.setdefault(m, MagicMock())
    sys.path.insert(0, REF)
    import mmrec
    from transformers.modeling_outputs import CausalLMOutputWithPast

    V, B, L = 160, 3, 24
    ANS, EOC, IMG, PAD, BOS = 150, 151, 152, 153, 154
    g = torch.Generator().manual_seed(7)

    def seq(n_chunks, n_pad, dangling):
        s = [BOS]
        for _ in range(n_chunks):
            s += [IMG] + torch.randint(0, 140, (2,), generator=g).tolist() + [ANS] + \
                 torch.randint(0, 140, (2,), generator=g).tolist() + [EOC]
        s += torch.randint(0, 140, (2,), generator=g).tolist() + [ANS] + torch.randint(0, 140, (1,), generator=g).tolist()
        if not dangling:
            s += [EOC]
        s = s[:L - n_pad]
        return s + [PAD] * (L - len(s))

    ids = torch.tensor([seq(2, 3, True), seq(1, 9, False), seq(3, 0, True)])
    # extra edge cases: <answer> as very first token, eoc without answer, answer right before pad
    ids[1, 0] = ANS
    weights = torch.tensor([2.0, 1.0, 2.0])
    base_logits = torch.randn(B, L, V, generator=g) * 2.0

    class Tok:
        pad_token_id = PAD

        def __call__(self, s, add_special_tokens=False):
            return {"input_ids": [{"<image>": IMG, "<|endofchunk|>": EOC, "<answer>": ANS}.get(s, 0)]}

    cap = {}

    class FakeLM(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.z = torch.nn.Parameter(base_logits.clone())

        def forward(self, vision_x, lang_x, attention_mask, labels):
            cap["labels"] = labels.clone()
            logits = self.z * 1.0
            logits.retain_grad()
            cap["logits"] = logits
            loss = torch.nn.functional.cross_entropy(logits[:, :-1].reshape(-1, V), labels[:, 1:].reshape(-1))
            return CausalLMOutputWithPast(loss=loss, logits=logits)

    class Acc:
        sync_gradients = True

        def accumulate(self, m):
            return contextlib.nullcontext()

        def backward(self, loss):
            cap["loss"] = loss.detach().clone()
            loss.backward()

        def clip_grad_norm_(self, params, mx):
            cap["clip_called"] = True

    for gamma in (0, 2):
        for rw in (0, 1):
            model = FakeLM()
            args = types.SimpleNamespace(num_epochs=1, precision="fp32", task="rec", gamma=gamma, rank=1,
                                         use_reweight=bool(rw), mask_lm_head=False, gradient_accumulation_steps=1,
                                         batch_size=B, world_size=1, report_to_wandb=False, logging_steps=1000)
            batch = {"net_input": {"patch_images": torch.zeros(B, 2, 3, 4, 4), "input_ids": ids.clone(),
                                   "attention_masks": (ids != PAD).long(), "weights": weights.clone()}}
            opt = torch.optim.SGD(model.parameters(), lr=0.0)
            sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)
            mmrec.train_one_epoch(args, model, 0, [batch], Tok(), opt, sched, 0, Acc(), MagicMock())
            np.savez_compressed(os.path.join(OUT, f"train_step_g{gamma}_rw{rw}.npz"),
                                ids=ids.numpy(), weights=weights.numpy(), logits=base_logits.numpy(),
                                special=np.array([ANS, EOC, PAD, IMG]), gamma=np.array(gamma), use_reweight=np.array(rw),
                                labels=cap["labels"].numpy(), loss=cap["loss"].numpy(),
                                dlogits=cap["logits"].grad.numpy())
            print("train_step", gamma, rw, float(cap["loss"]))


# ------------------------------------------------------------------ 2/3. in-tree xformers_model
def gen_intree():
    install_xformers_stub()
    sys.path.insert(0, REF)
    from transformers import CLIPVisionConfig, LlamaConfig
    from xformers_model.clip import CLIPVisionModel
    from xformers_model.llama import LlamaForCausalLM
    torch.manual_seed(3)
    cfg = CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                           image_size=32, patch_size=8, hidden_act="quick_gelu")
    m = CLIPVisionModel(cfg).eval()
    for p in m.parameters():
        p.data.normal_(0, 0.2)
    x = torch.randn(2, 3, 32, 32)
    with torch.no_grad():
        out = m(pixel_values=x)
    sd = {k: v for k, v in m.state_dict().items() if "position_ids" not in k}
    np.savez_compressed(os.path.join(OUT, "clip_tiny.npz"), pixels=x.numpy(),
                        last_hidden_state=out.last_hidden_state.numpy(), pooler_output=out.pooler_output.numpy(),
                        **{"sd." + k: v for k, v in _np(sd).items()})
    print("clip_tiny", tuple(out.last_hidden_state.shape))

    # head dim 64 variant for the HIP ViT (attention kernels: head dims 64 / 80 / 128), bf16-representable weights / pixels
    torch.manual_seed(8)
    cfg = CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                           image_size=32, patch_size=8, hidden_act="quick_gelu")
    m = CLIPVisionModel(cfg).eval()
    for p in m.parameters():
        p.data.normal_(0, 0.1)
        p.data = p.data.to(torch.bfloat16).float()
    x = torch.randn(3, 3, 32, 32).to(torch.bfloat16).float()
    with torch.no_grad():
        out = m(pixel_values=x)
    sd = {k: v for k, v in m.state_dict().items() if "position_ids" not in k}
    np.savez_compressed(os.path.join(OUT, "clip_hd64.npz"), pixels=x.numpy(),
                        last_hidden_state=out.last_hidden_state.numpy(), pooler_output=out.pooler_output.numpy(),
                        **{"sd." + k: v for k, v in _np(sd).items()})
    print("clip_hd64", tuple(out.last_hidden_state.shape))

    torch.manual_seed(4)
    lc = LlamaConfig(vocab_size=97, hidden_size=64, intermediate_size=112, num_hidden_layers=2,
                     num_attention_heads=4, max_position_embeddings=64, rms_norm_eps=1e-6, pad_token_id=0)
    lm = LlamaForCausalLM(lc).train()
    for n, p in lm.named_parameters():
        p.data.normal_(0, 0.15)
        if "norm" in n:
            p.data.add_(1.0)
    ids = torch.randint(1, 97, (2, 19))
    out = lm(input_ids=ids, attention_mask=torch.ones_like(ids), labels=ids)
    out.loss.backward()
    sd = {k: v for k, v in lm.state_dict().items() if "inv_freq" not in k}
    np.savez_compressed(os.path.join(OUT, "llama_tiny.npz"), ids=ids.numpy(), logits=out.logits.detach().numpy(),
                        loss=out.loss.detach().numpy(),
                        **{"sd." + k: v for k, v in _np(sd).items()},
                        **{"grad." + n: p.grad.numpy() for n, p in lm.named_parameters()})
    print("llama_tiny loss", float(out.loss))

    # head dim 64 variant (the HIP attention kernels support head dims 64 / 80 / 128), bf16-representable weights
    torch.manual_seed(6)
    lc = LlamaConfig(vocab_size=200, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                     num_attention_heads=2, max_position_embeddings=64, rms_norm_eps=1e-6, pad_token_id=0)
    lm = LlamaForCausalLM(lc).train()
    for n, p in lm.named_parameters():
        p.data.normal_(0, 0.08)
        if "norm" in n:
            p.data.add_(1.0)
        p.data = p.data.to(torch.bfloat16).float()
    ids = torch.randint(1, 200, (2, 40))
    out = lm(input_ids=ids, attention_mask=torch.ones_like(ids), labels=ids)
    out.loss.backward()
    sd = {k: v for k, v in lm.state_dict().items() if "inv_freq" not in k}
    np.savez_compressed(os.path.join(OUT, "llama_hd64.npz"), ids=ids.numpy(), logits=out.logits.detach().numpy(),
                        loss=out.loss.detach().numpy(),
                        **{"sd." + k: v for k, v in _np(sd).items()},
                        **{"grad." + n: p.grad.numpy() for n, p in lm.named_parameters()})
    print("llama_hd64 loss", float(out.loss))


# ------------------------------------------------------------------ 4. HF towers
def gen_hf_towers():
    from transformers import GPTNeoXConfig, GPTNeoXForCausalLM, OPTConfig, OPTForCausalLM
    torch.manual_seed(5)
    ids = torch.randint(3, 120, (2, 21))
    mask = torch.ones_like(ids)
    mask[0, 15:] = 0
    for par in (False, True):
        c = GPTNeoXConfig(vocab_size=128, hidden_size=80, num_hidden_layers=2, num_attention_heads=4,
                          intermediate_size=160, rotary_pct=1.0 if not par else 0.5, use_parallel_residual=par,
                          layer_norm_eps=1e-5, hidden_act="gelu", tie_word_embeddings=False, attn_implementation="eager")
        m = GPTNeoXForCausalLM(c).eval()
        for p in m.parameters():
            p.data.normal_(0, 0.2)
        with torch.no_grad():
            lg = m(input_ids=ids, attention_mask=mask).logits
        sd = {k.replace("lm_head.", "embed_out."): v for k, v in m.state_dict().items() if "inv_freq" not in k}
        np.savez_compressed(os.path.join(OUT, f"neox_tiny_par{int(par)}.npz"), ids=ids.numpy(), mask=mask.numpy(),
                            logits=lg.numpy(), **{"sd." + k: v for k, v in _np(sd).items()})
        print("neox_tiny", par, tuple(lg.shape))
    c = OPTConfig(vocab_size=128, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, ffn_dim=128,
                  max_position_embeddings=64, do_layer_norm_before=True, word_embed_proj_dim=64,
                  attn_implementation="eager")
    m = OPTForCausalLM(c).eval()
    for p in m.parameters():
        p.data.normal_(0, 0.2)
    with torch.no_grad():
        lg = m(input_ids=ids, attention_mask=mask).logits
    np.savez_compressed(os.path.join(OUT, "opt_tiny.npz"), ids=ids.numpy(), mask=mask.numpy(), logits=lg.numpy(),
                        **{"sd." + k: v for k, v in _np(m.state_dict()).items()})
    print("opt_tiny", tuple(lg.shape))
    from transformers import MptConfig, MptForCausalLM
    for heads in (4, 6):                               # 6: the interleaved slopes of a non-power-of-two head count
        m = MptForCausalLM(MptConfig(d_model=16 * heads, n_heads=heads, n_layers=2, expansion_ratio=4, max_seq_len=64, vocab_size=128,
                                     no_bias=True, layer_norm_epsilon=1e-5)).eval()
        for p in m.parameters():
            p.data.normal_(0, 0.2)
        with torch.no_grad():
            lg = m(input_ids=ids, attention_mask=mask).logits
        sd = {k: v for k, v in m.state_dict().items() if not k.startswith("lm_head")}
        np.savez_compressed(os.path.join(OUT, f"mpt_tiny_h{heads}.npz"), ids=ids.numpy(), mask=mask.numpy(), logits=lg.numpy(),
                            **{"sd." + k: v for k, v in _np(sd).items()})
        print("mpt_tiny", heads, tuple(lg.shape))


# ------------------------------------------------------------------ 5. checkpoint format (F3)
def gen_checkpoint():
    """the reference's OWN ``get_checkpoint`` (UniMP/pipeline/train/train_utils.py:258-265, what mmrec.py:873-892 saves) applied
    to the oracle Flamingo (module tree / parameter names of open_flamingo) after the factory's freezing:
    the key list, every tensor's shape and two small tensors -> checkpoint_keys.npz.  Note what the fixture pins: the
    function deletes the frozen names ``named_parameters()`` yields -- the FIRST path of a shared module -- so the duplicate
    paths (``old_decoder_blocks.*`` of the frozen LM blocks, ``gated_cross_attn_layers.*``) stay in the file."""
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from pipeline.train.train_utils import get_checkpoint            # the reference's function, unmodified
    import _parity as P
    m, layout = P.build_oracle(P.TINY)
    sd = get_checkpoint(m)
    keys = sorted(sd)
    pick = ["perceiver.latents", "lang_encoder.gated_cross_attn_layers.1.attn_gate"]
    np.savez_compressed(os.path.join(OUT, "checkpoint_keys.npz"), keys=np.array(keys),
                        shapes=np.array([",".join(map(str, sd[k].shape)) for k in keys]),
                        n_named_trainable=np.array(sum(1 for _, p in m.named_parameters() if p.requires_grad)),
                        **{"t." + k: sd[k].detach().numpy() for k in pick})
    print("checkpoint_keys", len(keys), "keys;", sum("old_decoder_blocks" in k for k in keys), "under old_decoder_blocks")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["hf", "intree", "ckpt", "train"]
    if "ckpt" in which:
        gen_checkpoint()
    if "hf" in which:
        gen_hf_towers()
    if "intree" in which:
        gen_intree()
    if "train" in which:
        gen_train_step()       # last: it MagicMocks a long list of modules
