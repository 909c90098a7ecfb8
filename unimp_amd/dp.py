"""Data-parallel gradient exchange: bucketed all-reduce(sum) over RCCL/xGMI, launched from autograd hooks
while backward is still running (replaces accelerate/DeepSpeed ZeRO-2 reduce-scatter, UniMP/mmrec.py:175,215,706-721).

One process per GPU; gradients live in FlatAdamW's contiguous bf16 buffer, so a bucket is a slice of it.
Buckets are cut in flat order (= reverse execution order, see optim.py); a bucket's all-reduce is issued
(async, on the communication stream RCCL owns) as soon as the last of its parameters has accumulated its
gradient.  The 1/world_size average is folded into the optimizer's ``grad_scale``; the clip norm is then
computed locally on identical reduced gradients, so no further collective is needed.
xGMI is point-to-point (7 links x ~153 GB/s): few large buckets (default 256 MiB) keep each ring step long
enough to be link-bandwidth- rather than latency-bound.
"""
import torch
import torch.distributed as dist


class GradBucketer:
    def __init__(self, optimizer, bucket_bytes=256 << 20, process_group=None, late_params=()):
        """late_params: parameters used more than once per step (tied embedding/head): their bucket is only
        reduced in finish(), because the first of their gradient accumulations does not mean the gradient is complete."""
        self.opt = optimizer
        self._late = {id(p) for p in late_params}
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.buckets = []        # [start, end, n_params]
        elems = max(1, bucket_bytes // optimizer.flat_g.element_size())
        cur = None
        self.param_bucket = {}
        for n, p, o, k in optimizer.layout:
            end = o + (k + optimizer.ALIGN - 1) // optimizer.ALIGN * optimizer.ALIGN
            if cur is None or (end - cur[0]) > elems and cur[2] > 0:
                cur = [o, end, 0]
                self.buckets.append(cur)
            cur[1] = end
            cur[2] += 1
            self.param_bucket[id(p)] = len(self.buckets) - 1
        self._pending = [b[2] for b in self.buckets]
        self._handles = []
        self._hooks = []
        self.sync = True          # False: gradient-accumulation micro-step, gradients only add up locally (accelerate's no_sync)
        if self.world > 1:
            for n, p, o, k in optimizer.layout:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _on_grad(self, p):
        if id(p) in self._late or not self.sync:
            return
        bi = self.param_bucket[id(p)]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def _launch(self, bi):
        s, e, _ = self.buckets[bi]
        self._handles.append(dist.all_reduce(self.opt.flat_g[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def finish(self):
        """call after backward: flush buckets whose params got no gradient this step, wait for all reductions."""
        if self.world > 1:
            self.opt._reattach()
            for bi, left in enumerate(self._pending):
                if left > 0 or not self.sync:      # accumulated micro-steps ran without exchange: reduce everything now
                    self._launch(bi)
            for h in self._handles:
                h.wait()
        self._handles = []
        self._pending = [b[2] for b in self.buckets]
        return 1.0 / self.world          # grad_scale for FlatAdamW.step

    def remove(self):
        for h in self._hooks:
            h.remove()
