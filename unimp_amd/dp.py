"""Data-parallel gradient exchange: bucketed all-reduce(sum) over RCCL/xGMI, launched from autograd hooks
while backward is still running (replaces accelerate/DeepSpeed ZeRO-2 reduce-scatter, UniMP/mmrec.py:175,215,706-721).

One process per GPU; gradients live in FlatAdamW's contiguous bf16 buffer, so a bucket is a slice of it.
Buckets are cut in flat order ([decay group | no-decay group], each in reverse execution order, see optim.py); a bucket
becomes READY when the last of its parameters has accumulated its gradient, and ready buckets are issued (async, on the
communication stream RCCL owns) **strictly in one static order**: by expected completion in backward, i.e. by the
registration index of the bucket's EARLIEST-registered parameter, latest first (the LM head's bucket, which backward
finishes first, leads although it sits in the no-decay half of the flat buffer; the Perceiver / input-embedding buckets
close).  A bucket goes out only after all its predecessors in that order have.  Every rank therefore issues the
same sequence of collectives whatever its autograd order or a parameter without a gradient on one rank does (RCCL, like
NCCL, pairs collectives by issue order; a bucket that never completes on some rank simply holds the later ones back until
finish(), on that rank only).  Buckets holding a ``late_params`` entry (a tied embedding / head: its first accumulation is
not its last) always go out in finish(), after all the others, in index order -- a static set, the same on every rank.
A parameter whose ``.grad`` is not the flat-buffer view any more (foreign code set it to None or replaced it) is folded
back into the view inside its hook, i.e. BEFORE its bucket can be issued.
The 1/world_size average is folded into the optimizer's ``grad_scale``; the clip norm is then computed locally on
identical reduced gradients, so no further collective is needed.
xGMI is point-to-point (7 links x ~153 GB/s): few large buckets (default 256 MiB) keep each ring step long
enough to be link-bandwidth- rather than latency-bound.
"""
import torch
import torch.distributed as dist


class GradBucketer:
    def __init__(self, optimizer, bucket_bytes=256 << 20, process_group=None, late_params=(), force_hooks=False):
        """late_params: parameters used more than once per step (tied embedding/head): their bucket is only
        reduced in finish(), because the first of their gradient accumulations does not mean the gradient is complete.
        force_hooks: register the hooks and run the collectives even when world_size == 1 (a 1-rank process group
        exercises the whole RCCL path -- init, async all_reduce on slices of the flat buffer, stream hand-off -- on a
        single GPU: tests/test_dp_gpu.py)."""
        self.opt = optimizer
        self._late = {id(p) for p in late_params}
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or ((force_hooks or getattr(optimizer, "shard", None) is not None) and dist.is_initialized())
        self.buckets = []        # [start, end, n_params]
        elems = max(1, bucket_bytes // optimizer.flat_g.element_size())
        cur = None
        self.param_bucket = {}
        self._slot = {}          # id(param) -> (offset, numel) of its view in the flat gradient buffer
        self.sharded = getattr(optimizer, "shard", None) is not None
        if self.sharded:         # the optimizer cut (and padded) the buckets itself: rank r owns the r-th 1/world of each
            self.buckets = [[s_, e_, 0] for s_, e_ in optimizer.buckets]
            bi = 0
            for n, p, o, k in optimizer.layout:
                while o >= self.buckets[bi][1]:
                    bi += 1
                self.buckets[bi][2] += 1
                self.param_bucket[id(p)] = bi
                self._slot[id(p)] = (o, k)
        else:
            for n, p, o, k in optimizer.layout:
                end = o + (k + optimizer.ALIGN - 1) // optimizer.ALIGN * optimizer.ALIGN
                if cur is None or (end - cur[0]) > elems and cur[2] > 0:
                    cur = [o, end, 0]
                    self.buckets.append(cur)
                cur[1] = end
                cur[2] += 1
                self.param_bucket[id(p)] = len(self.buckets) - 1
                self._slot[id(p)] = (o, k)
        self._late_buckets = sorted({self.param_bucket[i] for i in self._late if i in self.param_bucket})
        # static issue order during backward: expected completion = the bucket's earliest-registered parameter, latest first
        reg = getattr(optimizer, "reg_index", None) or {}
        first = [None] * len(self.buckets)
        for n, p, o, k in optimizer.layout:
            bi, r = self.param_bucket[id(p)], reg.get(id(p), 0)
            first[bi] = r if first[bi] is None else min(first[bi], r)
        self._order = sorted((bi for bi in range(len(self.buckets)) if bi not in self._late_buckets),
                             key=lambda bi: (-(first[bi] or 0), bi))
        self.last_launch_log = []
        self._reset()
        self._hooks = []
        self.sync = True          # False: gradient-accumulation micro-step, gradients only add up locally (accelerate's no_sync)
        self.launch_log = []      # bucket indices in issue order, last step (tests)
        self.record_exposed = False   # bench.py: bracket the waits of finish() with HIP events on the compute stream
        self.exposed_events = []      # (start, end) per step: the part of the exchange that backward did not hide
        self.record_ready = False     # bench.py: a HIP event on the compute stream at every bucket's issue point (when its gradients are complete)
        self.ready_events = []        # per step: (step-start event, [(bucket index, bytes, event)], finish event)
        self._step_ready = None
        if self.active:
            for n, p, o, k in optimizer.layout:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _reset(self):
        self._pending = [b[2] for b in self.buckets]
        self._next = 0            # position in self._order of the next bucket to issue
        self._issued = set()
        self._handles = []
        self._counted = set()     # parameters whose completion has been counted this step

    def _on_grad(self, p):
        o, k = self._slot[id(p)]
        view = self.opt.flat_g[o:o + k]
        g = p.grad
        if g is not None and g.data_ptr() != view.data_ptr():          # a stray gradient tensor: fold it in before any launch
            view.add_(g.reshape(-1).to(view.dtype))
            p.grad = view.view(p.shape)
        if id(p) in self._late or not self.sync:
            return
        # once per parameter and step.  A weight whose gradient the dW GEMM added straight into the flat buffer is reported by the
        # Trainer's sink (functional.WGRAD_SINK -> done()) AND, in the torch version this was written against, by autograd's
        # post-accumulate hook, which fires for the parameter although the Function returned no gradient for it
        if id(p) in self._counted:
            return
        self._counted.add(id(p))
        bi = self.param_bucket[id(p)]
        self._pending[bi] -= 1
        self._drain()

    def _drain(self):
        """issue every ready bucket at the head of the order; stop at the first one that is not complete."""
        while self._next < len(self._order) and self._pending[self._order[self._next]] <= 0:
            self._launch(self._order[self._next])
            self._next += 1

    def mark_step_start(self):
        """bench.py (record_ready): the compute-stream time origin of this step's bucket-ready timeline"""
        if self.record_ready and self.opt.flat_g.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._step_ready = (ev, [])

    def _launch(self, bi):
        s, e, _ = self.buckets[bi]
        self._issued.add(bi)
        self.launch_log.append(bi)
        g = self.opt.flat_g
        if self.record_ready and self._step_ready is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._step_ready[1].append((bi, (e - s) * g.element_size(), ev))
        if self.sharded and dist.get_backend(self.pg) == "nccl":
            # ZeRO-2: the rank only needs the sum over its own 1/world of the bucket (in place: output = its slice of the input)
            lo, hi, _ = self.opt.owned[bi]
            self._handles.append(dist.reduce_scatter_tensor(g[lo:hi], g[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        else:       # replicated state, or a backend without reduce-scatter (gloo in the tests): the owned slice of the sum is what is used
            self._handles.append(dist.all_reduce(g[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def finish(self):
        """call after backward: issue, in index order, the buckets that did not go out during backward (parameters
        without a gradient this step, late parameters, accumulation micro-steps), then wait for all reductions."""
        if self.active:
            self.opt._reattach()                     # only touches gradients of buckets that have NOT been issued (see _on_grad)
            for bi in self._order[self._next:]:
                self._launch(bi)
            for bi in self._late_buckets:
                self._launch(bi)
            # Work.wait() on a device tensor makes the compute stream wait for the communication stream (no host block):
            # the gap the compute stream spends there is the exposed part of the exchange
            ev = None
            if self.record_exposed and self.opt.flat_g.is_cuda:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            if self.record_ready and self._step_ready is not None:
                fe = torch.cuda.Event(enable_timing=True)
                fe.record()
                self.ready_events.append(self._step_ready + (fe,))
                self._step_ready = None
            for h in self._handles:
                h.wait()
            if ev is not None:
                ev[1].record()
                self.exposed_events.append(ev)
        log = self.launch_log
        self._reset()
        self.launch_log = []
        self.last_launch_log = log
        return 1.0 / self.world          # grad_scale for FlatAdamW.step

    def exposed_ms(self):
        """per-step exposed exchange time (ms) of the steps recorded so far; synchronises on the last event."""
        out = [a.elapsed_time(b) for a, b in self.exposed_events if (b.synchronize() or True)]
        return out

    def ready_timeline(self):
        """per recorded step: (ms from step start at which backward is over, [(bucket, bytes, ms from step start at which it could go out)])"""
        out = []
        for e0, lst, fe in self.ready_events:
            fe.synchronize()
            out.append((e0.elapsed_time(fe), [(bi, nb, e0.elapsed_time(ev)) for bi, nb, ev in lst]))
        return out

    @staticmethod
    def model_exposed_ms(timeline, world, bus_gbps, latency_us=30.0, sharded=False):
        """exchange model over a measured bucket-ready timeline: ring all-reduce of S bytes moves 2 (W - 1) / W x S per GPU, buckets go out in
        issue order on one communication stream at `bus_gbps` (RCCL's 'bus bandwidth') + a fixed latency each; returns the ms the compute
        stream would wait in finish() -- what backward does not hide.  sharded (ZeRO-2 layout, Trainer(shard_optimizer=True)): the buckets
        go out as reduce-scatters ((W - 1) / W x S per GPU: half the all-reduce's bytes) and the function returns a pair: (exposed
        reduce-scatter ms, ms of the parameter all-gather that follows the sharded update -- (W - 1) / W x S per GPU per bucket, nothing of
        the step left to hide it behind: the next forward's first trainable read waits for it)."""
        t_end, lst = timeline
        free = 0.0
        f = (1.0 if sharded else 2.0) * (world - 1) / world
        for bi, nb, t_ready in lst:
            start = max(free, t_ready)
            free = start + latency_us * 1e-3 + f * nb / (bus_gbps * 1e9) * 1e3
        exposed = max(0.0, free - t_end)
        if not sharded:
            return exposed
        gather = sum(latency_us * 1e-3 + (world - 1) / world * nb / (bus_gbps * 1e9) * 1e3 for _, nb, _ in lst)
        return exposed, gather

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        cb, self.on_remove = getattr(self, "on_remove", None), None
        if cb is not None:
            cb()
