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
                 device=None, shard=None):
        """shard = (rank, world, bucket_elems, process_group): ZeRO-2-style optimizer-state sharding (the reference trains under
        DeepSpeed ZeRO-2: accelerate_config_zero2.yaml:2-9).  The flat buffers are then cut into buckets of ~bucket_elems
        elements, each padded to a multiple of world x ALIGN, and rank r owns the r-th 1/world of EVERY bucket: fp32 master / m /
        v exist for the owned slices only (12 B/param / world), gradients arrive by reduce-scatter (dp.GradBucketer), the clip
        norm is the all-reduced sum of the owned slices' squares, and the updated bf16 parameters are all-gathered bucket by
        bucket.  None (default): replicated state -- at 4b-instruct 16 GB of 288, sharding buys nothing there; at 9B it frees
        ~15 GB per GPU."""
        named = [(n, p) for n, p in named_parameters if p.requires_grad]
        if not named:
            raise ValueError("no trainable parameters")
        seen, uniq = set(), []
        for n, p in named:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append((n, p))
        self.reg_index = {id(p): i for i, (n, p) in enumerate(uniq)}      # registration (~ forward execution) order, for dp.GradBucketer
        decay = [(n, p) for n, p in uniq if apply_decay(n)][::-1]
        nodecay = [(n, p) for n, p in uniq if not apply_decay(n)][::-1]
        self.lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.step_count = 0
        dev = device or uniq[0][1].device
        self.layout = []          # (name, param, offset, numel)
        self.shard = shard
        self.buckets = None       # sharded: [(start, end)] of the flat buffers, each (end - start) % (world * ALIGN) == 0
        off = 0
        rnd = lambda k: (k + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        bstart, buckets = 0, []
        quantum = shard[1] * self.ALIGN if shard else 0

        def place(n, p):
            nonlocal off, bstart
            if shard and off > bstart and off + rnd(p.numel()) - bstart > shard[2]:      # close the bucket before this parameter
                off = bstart + (off - bstart + quantum - 1) // quantum * quantum
                buckets.append((bstart, off))
                bstart = off
            self.layout.append((n, p, off, p.numel()))
            off += rnd(p.numel())
        for n, p in decay:
            place(n, p)
        self.n_decay = off       # decay applies to flat indices below this (padding elements hold zeros and stay zero)
        for n, p in nodecay:
            place(n, p)
        if shard:
            off = bstart + (off - bstart + quantum - 1) // quantum * quantum
            buckets.append((bstart, off))
            self.buckets = buckets
            # a bucket may straddle the decay boundary only at a parameter boundary: n_decay is a parameter offset, fine
        self.total = off
        self.flat_p = torch.zeros(off, dtype=torch.bfloat16, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.bfloat16, device=dev)
        if shard:
            r, w = shard[0], shard[1]
            self.owned = []           # (global lo, global hi, offset in the state buffers) per bucket
            so = 0
            for s_, e_ in buckets:
                c = (e_ - s_) // w
                self.owned.append((s_ + r * c, s_ + (r + 1) * c, so))
                so += c
            n_state = so
        else:
            self.owned = [(0, off, 0)]
            n_state = off
        self.master = torch.zeros(n_state, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n_state, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n_state, dtype=torch.float32, device=dev)
        self.norm_buf = torch.zeros(1 + 1024, dtype=torch.float32, device=dev)
        for n, p, o, k in self.layout:
            self.flat_p[o:o + k].copy_(p.detach().reshape(-1))
        if shard:
            # fp32 master of the owned slices from the original (possibly higher-precision) parameters
            full = torch.zeros(off, dtype=torch.float32, device=dev)
            for n, p, o, k in self.layout:
                full[o:o + k].copy_(p.detach().reshape(-1))
            for lo, hi, so in self.owned:
                self.master[so:so + hi - lo].copy_(full[lo:hi])
            del full
        else:
            for n, p, o, k in self.layout:
                self.master[o:o + k].copy_(p.detach().reshape(-1))
        for n, p, o, k in self.layout:
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
        b1, b2 = self.betas
        lr_ = self.lr if lr is None else lr
        if self.shard is None:
            ops.sumsq(self.flat_g, self.norm_buf)
            ops.adamw_flat(self.master, self.m, self.v, self.flat_p, self.flat_g, self.n_decay, lr_, b1, b2,
                           self.eps, self.weight_decay, self.step_count, self.norm_buf, grad_scale, self.max_grad_norm or 0.0, True)
            return
        import torch.distributed as dist
        pg = self.shard[3]
        # the owned slices hold the reduce-scattered (summed) gradient: their squares add up, over the ranks, to the global norm
        for lo, hi, so in self.owned:
            ops.sumsq(self.flat_g[lo:hi], self.norm_buf)
        dist.all_reduce(self.norm_buf[0:1], op=dist.ReduceOp.SUM, group=pg)
        for lo, hi, so in self.owned:
            k = hi - lo
            ops.adamw_flat(self.master[so:so + k], self.m[so:so + k], self.v[so:so + k], self.flat_p[lo:hi], self.flat_g[lo:hi],
                           max(0, min(self.n_decay - lo, k)), lr_, b1, b2, self.eps, self.weight_decay, self.step_count, self.norm_buf,
                           grad_scale, self.max_grad_norm or 0.0, False)
        self.flat_g.zero_()
        # every rank now holds fresh bf16 values for its slices: all-gather them bucket by bucket, in place
        nccl = dist.get_backend(pg) == "nccl"
        for (s_, e_), (lo, hi, so) in zip(self.buckets, self.owned):
            if nccl:
                dist.all_gather_into_tensor(self.flat_p[s_:e_], self.flat_p[lo:hi], group=pg)
            else:       # gloo (tests): list form through a temporary
                parts = [torch.empty_like(self.flat_p[lo:hi]) for _ in range(self.shard[1])]
                dist.all_gather(parts, self.flat_p[lo:hi].contiguous(), group=pg)
                self.flat_p[s_:e_].copy_(torch.cat(parts))

    def grad_norm(self):
        """device scalar: global L2 norm of the gradient the last step() clipped -- the averaged one, i.e. the summed
        buffer times grad_scale (1 / world / grad_accum) -- the value ``clip_grad_norm_`` returns in the reference
        (mmrec.py:247-248)."""
        return self.norm_buf[0].sqrt() * getattr(self, "_gscale", 1.0)

    def refresh_master(self):
        """fp32 master copies re-derived from the (bf16) parameters -- after weights were loaded into the model."""
        for lo, hi, so in self.owned:
            self.master[so:so + hi - lo].copy_(self.flat_p[lo:hi])

    def reset_state(self):
        """forget the moments and the step count (weights loaded without their optimizer state)."""
        self.m.zero_(); self.v.zero_()
        self.step_count = 0

    def _gather_state(self, buf):
        """full fp32 flat buffer (this optimizer's layout) of a sharded state buffer: every rank's owned slices, bucket by bucket.
        Collective: every rank of the group must call it."""
        import torch.distributed as dist
        pg, w = self.shard[3], self.shard[1]
        full = torch.zeros(self.total, dtype=torch.float32, device=buf.device)
        nccl = dist.get_backend(pg) == "nccl"
        for (s_, e_), (lo, hi, so) in zip(self.buckets, self.owned):
            mine = buf[so:so + hi - lo].contiguous()
            if nccl:
                dist.all_gather_into_tensor(full[s_:e_], mine, group=pg)
            else:
                parts = [torch.empty_like(mine) for _ in range(w)]
                dist.all_gather(parts, mine, group=pg)
                full[s_:e_].copy_(torch.cat(parts))
        return full

    def state_dict(self, to_host=True):
        """{"step", "lr", "names", "master" / "m" / "v": {parameter name: fp32 tensor}} -- per parameter, so a state saved by a
        replicated optimizer loads into a sharded one and back, whatever the bucket padding.  With a sharded state this is a
        COLLECTIVE (every rank gathers every slice, one state buffer at a time: a transient fp32 copy of the flat buffer);
        ``to_host=False`` takes part in the gathers but builds no host copies (the ranks that do not write the file)."""
        out = {"step": self.step_count, "lr": self.lr, "names": [n for n, _, _, _ in self.layout]}
        for key, buf in (("master", self.master), ("m", self.m), ("v", self.v)):
            full = buf if self.shard is None else self._gather_state(buf)
            if to_host:
                out[key] = {n: full[o:o + k].detach().to("cpu", copy=True) for n, _, o, k in self.layout}
            del full
        return out if to_host else None

    def load_state_dict(self, sd):
        assert sd["names"] == [n for n, _, _, _ in self.layout], "parameter layout changed"
        self.step_count = sd["step"]
        for key, buf in (("master", self.master), ("m", self.m), ("v", self.v)):
            src = sd[key]
            if torch.is_tensor(src):                      # round-1 files: one flat tensor in the replicated layout
                if self.shard is not None:
                    raise ValueError("a flat (round-1) optimizer state cannot be loaded into a sharded optimizer")
                buf.copy_(src)
                continue
            if self.shard is None:
                for n, _, o, k in self.layout:
                    buf[o:o + k].copy_(src[n].reshape(-1))
            else:                                         # every rank reads the whole file and keeps its slices
                full = torch.zeros(self.total, dtype=torch.float32, device=buf.device)
                for n, _, o, k in self.layout:
                    full[o:o + k].copy_(src[n].reshape(-1))
                for lo, hi, so in self.owned:
                    buf[so:so + hi - lo].copy_(full[lo:hi])
                del full
        if self.shard is not None:                        # bf16 parameters of every slice from the file's masters (identical on all ranks)
            for n, p, o, k in self.layout:
                self.flat_p[o:o + k].copy_(sd["master"][n].reshape(-1).to(self.flat_p.device))
            return
        self.flat_p.copy_(self.master.to(torch.bfloat16))
