"""Fused flat AdamW + global-norm clip for the trainable subset (UniMP/mmrec.py:247-256,609-631,671).

All trainable parameters are re-homed into ONE contiguous bf16 buffer (their ``.data`` become views) with a
matching bf16 gradient buffer (``.grad`` views) and fp32 master / m / v buffers: one kernel launch updates
everything, one reduction gives the clip norm, and data-parallel buckets are plain slices (dp.py).
Layout: [ weight-decay group | no-decay group ], each in reverse registration order so that gradients that
finish first in backward (late layers) sit first -- buckets then complete front to back.
Decay grouping reproduces the reference filter, including its ``ff.0.weight`` quirk.
"""
import math
import torch

from . import ops


def apply_decay(name):
    """mmrec.py:612-619"""
    return ("gated_cross_attn_layer" in name and "ff_gate" not in name and "attn_gate" not in name
            and "norm" not in name and "bias" not in name)


def cosine_lr(step, base_lr, warmup_steps, total_steps, num_cycles=0.5):
    """transformers.get_cosine_schedule_with_warmup (mmrec.py:687-693)"""
    if step < warmup_steps:
        return base_lr * float(step) / float(max(1, warmup_steps))
    prog = float(step - warmup_steps) / float(max(1, total_steps - warmup_steps))
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * num_cycles * 2.0 * prog)))


def linear_lr(step, base_lr, warmup_steps, total_steps):
    if step < warmup_steps:
        return base_lr * float(step) / float(max(1, warmup_steps))
    return base_lr * max(0.0, float(total_steps - step) / float(max(1, total_steps - warmup_steps)))


class FlatAdamW:
    ALIGN = 64   # elements; keeps every parameter view 128-byte aligned

    def __init__(self, named_parameters, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1, max_grad_norm=1.0,
                 device=None):
        named = [(n, p) for n, p in named_parameters if p.requires_grad]
        if not named:
            raise ValueError("no trainable parameters")
        seen, uniq = set(), []
        for n, p in named:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append((n, p))
        decay = [(n, p) for n, p in uniq if apply_decay(n)][::-1]
        nodecay = [(n, p) for n, p in uniq if not apply_decay(n)][::-1]
        self.lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.step_count = 0
        dev = device or uniq[0][1].device
        self.layout = []          # (name, param, offset, numel)
        off = 0
        rnd = lambda k: (k + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        for n, p in decay:
            self.layout.append((n, p, off, p.numel()))
            off += rnd(p.numel())
        self.n_decay = off
        for n, p in nodecay:
            self.layout.append((n, p, off, p.numel()))
            off += rnd(p.numel())
        self.total = off
        self.flat_p = torch.zeros(off, dtype=torch.bfloat16, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.bfloat16, device=dev)
        self.master = torch.zeros(off, dtype=torch.float32, device=dev)
        self.m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.v = torch.zeros(off, dtype=torch.float32, device=dev)
        self.norm_buf = torch.zeros(1 + 1024, dtype=torch.float32, device=dev)
        for n, p, o, k in self.layout:
            self.master[o:o + k].copy_(p.detach().reshape(-1))
            self.flat_p[o:o + k].copy_(p.detach().reshape(-1))
            p.data = self.flat_p[o:o + k].view(p.shape)
            p.grad = self.flat_g[o:o + k].view(p.shape)

    def zero_grad(self, set_to_none=False):
        self.flat_g.zero_()

    def _reattach(self):
        """autograd may have replaced .grad views (e.g. set_to_none by foreign code): fold such grads back in."""
        for n, p, o, k in self.layout:
            g = p.grad
            view = self.flat_g[o:o + k]
            if g is None:
                p.grad = view.view(p.shape)
            elif g.data_ptr() != view.data_ptr():
                view.add_(g.reshape(-1).to(view.dtype))
                p.grad = view.view(p.shape)

    def step(self, lr=None, grad_scale=1.0):
        """clip (global L2 norm of grad*grad_scale to max_grad_norm) + AdamW; zeroes the gradient buffer."""
        self._reattach()
        self.step_count += 1
        self._gscale = float(grad_scale)
        self.norm_buf[0:1].zero_()
        ops.sumsq(self.flat_g, self.norm_buf)
        b1, b2 = self.betas
        ops.adamw_flat(self.master, self.m, self.v, self.flat_p, self.flat_g, self.n_decay, self.lr if lr is None else lr, b1, b2,
                       self.eps, self.weight_decay, self.step_count, self.norm_buf, grad_scale, self.max_grad_norm or 0.0, True)

    def grad_norm(self):
        """device scalar: global L2 norm of the gradient the last step() clipped -- the averaged one, i.e. the summed
        buffer times grad_scale (1 / world / grad_accum) -- the value ``clip_grad_norm_`` returns in the reference
        (mmrec.py:247-248)."""
        return self.norm_buf[0].sqrt() * getattr(self, "_gscale", 1.0)

    def reset_state(self):
        """forget the moments and the step count (weights loaded without their optimizer state)."""
        self.m.zero_(); self.v.zero_()
        self.step_count = 0

    def state_dict(self):
        return {"step": self.step_count, "master": self.master, "m": self.m, "v": self.v,
                "names": [n for n, _, _, _ in self.layout], "lr": self.lr}

    def load_state_dict(self, sd):
        assert sd["names"] == [n for n, _, _, _ in self.layout], "parameter layout changed"
        self.step_count = sd["step"]
        self.master.copy_(sd["master"]); self.m.copy_(sd["m"]); self.v.copy_(sd["v"])
        self.flat_p.copy_(self.master.to(torch.bfloat16))
