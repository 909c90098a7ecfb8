"""One optimizer step with train_one_epoch's semantics (UniMP/mmrec.py:65-302), every stage on the HIP kernels:

  labels   = label-mask kernel                      (mmrec.py:143-168, Python double loop in the reference)
  output   = model(vision_x, lang_x, mask, labels)   (mmrec.py:177-181)
  loss     = weighted focal CE on output["logits"]   (mmrec.py:190-213)
  backward + bucketed RCCL all-reduce                (mmrec.py:215; dp.py)
  clip 1.0 + AdamW + LR schedule                     (mmrec.py:247-256; optim.py)

Loss normalisation is per rank and gradients are averaged across ranks (mean of per-rank token means), as
in the reference.  Samples/s accounting = GA * batch * world / step_time (mmrec.py:267-272).
"""
import os
import time
import torch

from . import functional as F_
from . import ops
from .optim import FlatAdamW, cosine_lr, linear_lr
from .dp import GradBucketer


class AverageMeter:
    """UniMP/pipeline/train/train_utils.py:268-284"""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def _wait_params(model):
    """an optimizer update still running on a Trainer's side stream (overlap_optimizer) lands before the current stream reads or
    writes a parameter: the model carries the update's event (``Flamingo._params_barrier``; a model without one has nothing pending)"""
    pb = getattr(model, "_params_barrier", None)
    if pb is not None:
        pb()


def get_checkpoint(model):
    """trainable-only state dict (train_utils.py:258-265)."""
    _wait_params(model)
    sd = model.state_dict()
    for name, p in model.named_parameters():
        if not p.requires_grad:
            del sd[name]
    return sd


def save_checkpoint(path, model, trainer=None, epoch=0, barrier=False, group=None):
    """UniMP writes only the trainable tensors (mmrec.py:873-892: ``get_checkpoint``) and never the optimizer, so its
    resume path is broken.  Same file format for the weights -- a flat ``{name: tensor}`` dict that
    ``model.load_state_dict(sd, strict=False)`` consumes -- plus, optionally, a side file with the fp32 optimizer state
    and the scheduler position so a run can actually resume.

    Calling convention under data parallelism:
      * replicated optimizer state (default): NOT a collective.  Rank 0 writes, every other rank returns at once without
        building any host copy -- the reference's pattern (mmrec.py:772-873: ``wait_for_everyone()`` then ``if args.rank == 0:``
        save) ports as is.  ``barrier=True`` makes it a collective that every rank calls and that returns once the file exists
        (save-then-load-on-every-rank patterns) -- without it a rank other than 0 may race rank 0's ``torch.save``.
      * sharded optimizer state (``Trainer(shard_optimizer=True)``) with a trainer given: a COLLECTIVE -- **every rank must
        call it**: the owned slices are gathered bucket by bucket, rank 0 keeps the host copy and writes, and a barrier
        closes the call so no rank runs ahead into the next step's collectives."""
    import torch.distributed as dist
    if trainer is not None:
        trainer.sync()                                 # an optimizer update on the side stream (overlap_optimizer) lands first
    else:
        _wait_params(model)                            # ... also when the caller did not pass its trainer
    if group is None and trainer is not None:
        group = trainer.dp.pg                          # the Trainer may run on a subgroup: WORLD would wait for ranks that never call
    rank0 = not dist.is_initialized() or dist.get_rank(group) == 0
    sharded = trainer is not None and trainer.opt.shard is not None
    if not rank0 and not sharded:
        if barrier and dist.is_initialized():
            dist.barrier(group=group)
        return
    o = None
    if trainer is not None:
        o = trainer.opt.state_dict(to_host=rank0)      # sharded: a collective; only rank 0 keeps the per-parameter host copies
    err = None
    if rank0:                                          # replicas hold identical weights: one writer
        try:
            sd = {k: v.detach().to("cpu") for k, v in get_checkpoint(model).items()}
            torch.save(sd, path)
            if o is not None:
                torch.save({"epoch": epoch, "sched_step": trainer.sched_step, "optimizer": o}, path + ".resume")
        except BaseException as e:                     # the barrier below must still be reached, or every other rank hangs in it
            err = e
    if sharded:
        dist.barrier(group=trainer.opt.shard[3])
    elif barrier and dist.is_initialized():
        dist.barrier(group=group)
    if err is not None:
        raise err


def load_checkpoint(path, model, trainer=None):
    """inverse of save_checkpoint; also accepts OpenFlamingo ``checkpoint.pt`` / UniMP ``weights_epoch_*.pt`` files
    (same parameter names).  Returns the epoch to resume from (0 without a .resume side file)."""
    import os
    # the side-stream update (overlap_optimizer) may still be writing flat_p / master: it lands before load_state_dict and
    # refresh_master touch them on the current stream
    if trainer is not None:
        trainer.sync()
    _wait_params(model)
    sd = torch.load(path, map_location="cpu")
    if "model_state_dict" in sd:
        sd = sd["model_state_dict"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if unexpected:
        raise KeyError(f"unexpected keys in checkpoint: {unexpected[:5]} ...")
    epoch = 0
    if trainer is not None:
        # parameters are views of the flat bf16 buffer: refresh the fp32 master copy from what was just loaded
        trainer.opt.refresh_master()
        if os.path.exists(path + ".resume"):
            r = torch.load(path + ".resume", map_location="cpu")
            trainer.opt.load_state_dict(r["optimizer"])
            trainer.sched_step, epoch = r["sched_step"], r["epoch"] + 1
        else:
            # weights only (the reference's own files): stale moments of the live trainer would belong to other weights
            trainer.opt.reset_state()
    return epoch


class _WgradSink:
    """functional.WGRAD_SINK of a Trainer: 2-D parameters -> their views of the optimizer's flat bf16 gradient buffer.
    Every weight that reaches the sink (the dense layers of the gated blocks and the Perceiver: LinearFn, MLPBlockFn,
    GatedXAttnFn, PerceiverAttnFn) is used ONCE per forward, so its gradient is complete after its one dW GEMM and ``done`` can
    tell the data-parallel bucketer so, exactly as autograd's post-accumulate hook would; a second use inside one backward is
    refused loudly instead of being reduced half-summed.  Tied / masked embeddings (``exclude``) never come here."""

    def __init__(self, opt, exclude=(), dp=None):
        ex = {id(p) for p in exclude}
        self.views = {}
        self.dp = dp
        self.seen = set()
        if opt.flat_g.dtype == torch.bfloat16:
            for n, p, o, k in opt.layout:
                if p.dim() in (1, 2) and p.numel() >= 8 and id(p) not in ex:      # dense weights; LayerNorm gamma / beta (functional._ln_bwd)
                    self.views[p.data_ptr()] = (p, opt.flat_g[o:o + k].view(p.shape))

    def has(self, w):
        """would view_of(w) hand out a view?  (no bookkeeping)"""
        e = None if w is None else self.views.get(w.data_ptr())
        if e is None or e[0].shape != w.shape:
            return False
        g = e[0].grad
        return g is not None and g.data_ptr() == e[1].data_ptr()

    def view_of(self, w):
        e = self.views.get(w.data_ptr())
        if e is None or e[0].shape != w.shape:
            return None
        p, v = e
        g = p.grad
        if g is None or g.data_ptr() != v.data_ptr():     # foreign code replaced .grad: autograd's path, folded back by _reattach
            return None
        if self.dp is not None and self.dp.sync:
            if id(p) in self.seen:
                raise RuntimeError("a weight routed through the weight-gradient sink was used twice in one backward under data "
                                   "parallelism: its bucket may already be on the wire (Trainer(direct_wgrad=False) disables the sink)")
            self.seen.add(id(p))
        return v

    def done(self, w):
        if self.dp is not None:
            self.dp._on_grad(self.views[w.data_ptr()][0])


_ROWS_SYNC = os.environ.get("UNIMP_ROWS_SYNC", "0") == "1"      # A/B knob: the blocking nonzero() of rounds 1-3
# fuse_accum=None (auto): fused accumulation holds the activations of GA x B samples at once, so the automatic choice takes it only
# while GA x B x L stays within this many tokens (b = 64 x L = 512, the bench's headline shape, is 32 768 tokens and needs ~150 GB at
# cfg2); beyond it the micro-batches run one after the other as the reference does.  fuse_accum=True is never second-guessed.
FUSE_TOKEN_BUDGET = int(os.environ.get("UNIMP_FUSE_TOKENS", str(48 * 1024)))


class StepOut(tuple):
    """what ``Trainer.step`` returns: unpacks as ``(loss, stats)`` like before, and carries ``pending`` -- True when the call only
    BUFFERED its micro-batch (fused accumulation before the GA-th micro-batch): no forward ran, ``loss`` / ``stats`` are then the
    previous optimizer step's values (NaN / zeros before the first one), not this micro-batch's.  A logging loop ported from
    mmrec.py:259-296 must skip pending results (``if not out.pending: meter.update(...)``)."""
    pending = False

    def __new__(cls, loss, stats, pending=False):
        t = super().__new__(cls, (loss, stats))
        t.pending = bool(pending)
        return t



class Trainer:
    def __init__(self, model, special_ids, lr=2e-4, weight_decay=0.1, gamma=2.0, use_reweight=True, max_grad_norm=1.0,
                 lr_scheduler="cosine", warmup_steps=0, total_steps=1000, bucket_bytes=256 << 20, process_group=None,
                 sparse_head=False, grad_accum=1, mask_lm_head=False, force_dp_hooks=False, dense_head_backward=False,
                 shard_optimizer=False, direct_wgrad=True, graph=False, fuse_accum=None, packed=None, overlap_optimizer=False):
        """packed (None: the UNIMP_PACKED environment default, off): packed token order in the language tower -- LayerNorm and the
        QKV / out / MLP / gated feed-forward projections run on the VALID tokens only ([1, M, H], M = the valid count rounded up to 2048
        rows), the attention kernels take the sequences as row ranges of the packed buffers (functional.Pack, include/unimp_hip.h
        q_row_off / k_row_off), the rotation reads each row's position from a table.  Needs right-padded sequences.  The reference
        computes the <PAD> rows too (collate_rec.py:38-74 pads to the longest sequence of the batch); nothing reads them: loss and
        gradients are those of the padded run (tests/test_model_gpu.py::test_packed_token_order_equals_padded), logits at <PAD>
        positions become those of a zero hidden state.  GPT-NeoX, MPT and OPT towers; one extra host sync per step (the valid count).
        fuse_accum (None = automatic: ON when grad_accum > 1, graph is off and GA x B x L of the first micro-batch fits
        ``FUSE_TOKEN_BUDGET`` tokens -- activation memory scales with GA x B, beyond the budget the micro-steps run one after the
        other; True = always; False = the sequential micro-steps of the reference): run the GA
        micro-batches of an optimizer step as ONE forward / backward pass.  The reference accumulates because 3 samples are what fits its GPUs (unimp_task.sh:2-30: --batch_size 3,
        --gradient_accumulation_steps 2); on 288 GB the activations of all GA micro-batches fit, and one pass over GA x B samples
        fills the GEMM tiles GA times better (a micro-batch of 3 x 512 tokens is 6 tile rows: measured, the step is GPU-bound at
        0.57 PFLOP/s in the GEMMs, not launch-bound).  SAME optimizer step: the loss keeps its per-micro-batch normalisation
        (mmrec.py:213 divides by the labeled positions of the micro-batch, accelerate averages the GA losses) through per-sample
        weights w_b * N_total / (GA * N_mb(b)) computed on the device -- gradients equal the sequential ones up to summation order
        (tests/test_model_gpu.py::test_fused_accumulation_equals_sequential).  CONTRACT CHANGE against the sequential loop: ``step()`` only
        BUFFERS the first GA - 1 micro-batches of an optimizer step -- it returns a ``StepOut`` with ``pending=True`` whose loss / stats are
        the PREVIOUS optimizer step's (a NaN loss and zero stats before the first one: a meter that averages them without looking at
        ``pending`` goes NaN instead of being silently skewed); the GA-th call runs the fused pass and returns its loss.  Micro-batches of
        different lengths are right-padded.  When the loader ends inside a group, call ``flush()`` (accelerate steps at the end of
        the dataloader: ``sync_with_dataloader``) -- otherwise the buffered micro-batches would join the next epoch's first group.
        overlap_optimizer (off by default): clip + AdamW (HBM-bound: 28 B per trainable parameter, 7.3 ms at cfg2) run on a SECOND stream
        and the next step's forward starts at once on the main stream -- the ViT is frozen and runs under no_grad, so nothing it reads or
        writes is touched by the update; ``Flamingo._encode_vision_x`` makes the main stream wait for the update right before the
        Perceiver, the first reader of a trainable parameter (``model._params_ready``).  Same kernels, same order per buffer: the
        parameters equal the serial order's BIT FOR BIT (tests/test_model_gpu.py::test_overlapped_optimizer_equals_serial).  At the
        reference's b = 3 x GA 2 the update is 8.6 % of the step, at b = 64 about 1 %.  Needs a frozen vision encoder and replicated
        optimizer state; code that reads parameters or optimizer buffers between steps (checkpoints do it themselves) calls ``sync()``.
        graph (off by default): replay the forward + loss + backward of a micro-batch as ONE HIP graph.  At the reference's
        shipped shape (--batch 3 --grad-accum 2, unimp_task.sh:2-30) a micro-step is ~3 000 launches of kernels that run for
        10-40 us each: the host, not the GPU, sets the pace.  The first micro-step with a given set of batch shapes runs eagerly
        (it warms the autotuner, the frozen-weight caches and the allocator), the second is captured (inputs copied to static
        buffers), every later one is a copy + replay.  Needs the sync-free loss path, so it implies ``dense_head_backward``; the
        returned loss / stats are static tensors that the next replay overwrites; under data parallelism the gradient exchange
        of a graphed step is issued from ``finish()`` (nothing Python-side runs during a replay), i.e. not overlapped.
        sparse_head (off by default): apply the LM head and the loss only to the positions whose next token carries a
        label -- identical loss / gradients / update (unlabeled rows contribute nothing), ~5 % fewer FLOPs at cfg2; the
        returned model output then has no logits.  Costs one host sync per step (the row count).
        dense_head_backward (off by default): form the dense [B*L, V] logit gradient and run the head's dX / dW GEMMs over all
        B*L rows as the reference does.  The default computes the SAME dense forward logits and loss but restricts the head's
        backward to the positions that carry a label -- the gradient is exactly zero on the others (functional.DenseHeadLossFn);
        it needs the number of labeled positions on the host: one sync per step, right after the label-mask kernel."""
        self.model, self.sparse_head, self.dense_head_backward = model, sparse_head, dense_head_backward or graph
        if graph and sparse_head:
            raise ValueError("Trainer(graph=True) needs the sync-free dense loss path (sparse_head takes a row count on the host)")
        self.use_graph, self._graph = graph, None
        if fuse_accum and graph:
            raise ValueError("Trainer(fuse_accum=True, graph=True): the fused pass is not graph-captured (it would silently ignore graph=True); "
                             "pick one -- graph=True replays each micro-batch, fuse_accum=True runs the GA micro-batches as one pass")
        # None: decided at the first micro-batch, when B and L are known (_fuse_decide)
        self._fuse_auto = fuse_accum is None and grad_accum > 1 and not graph
        self.fuse_accum, self._stash, self._last = bool(fuse_accum and grad_accum > 1), [], None
        self._mb_index, self._cnt_host = {}, None
        le_ = model.lang_encoder
        if packed is not None:                   # per tower, not the module-level default: a second Trainer leaves this one alone
            if packed and not getattr(le_, "supports_packed", True):
                raise ValueError(f"Trainer(packed=True): {type(le_).__name__} has no packed-row form (QK-LayerNorm / Llama towers); use packed=False")
            le_.packed = bool(packed)
        self.packed = bool(F_.PACKED if getattr(le_, "packed", None) is None else le_.packed) and getattr(le_, "supports_packed", True)
        if graph and self.packed:
            raise ValueError("Trainer(graph=True) cannot be combined with the packed token order (its valid-token count is a host sync per step)")
        self.grad_accum, self._micro = grad_accum, 0       # mmrec.py's --gradient_accumulation_steps (accelerator.accumulate)
        self.ids = special_ids                   # dict(answer_id, eoc_id, pad_id, media_id)
        self.gamma, self.use_reweight = gamma, use_reweight
        shard = None
        if shard_optimizer:       # ZeRO-2-style optimizer-state sharding over the data-parallel group (optim.FlatAdamW)
            import torch.distributed as dist
            if not dist.is_initialized():
                raise RuntimeError("shard_optimizer needs an initialised process group")
            shard = (dist.get_rank(process_group), dist.get_world_size(process_group), max(1, bucket_bytes // 2), process_group)
        self.opt = FlatAdamW(model.named_parameters(), lr=lr, weight_decay=weight_decay, max_grad_norm=max_grad_norm, shard=shard)
        le = model.lang_encoder
        late = [le.get_input_embeddings().weight] if getattr(le, "tied", False) else []
        # --mask_lm_head (mmrec.py:218-229): only the <answer> row of the embedding / head gradients survives.  The rows are
        # cleared after backward, so these parameters' buckets must not be exchanged from the hooks: they go out in finish()
        self._masked = []
        if mask_lm_head:
            self._masked = [w for w in {id(m.weight): m.weight for m in (le.get_input_embeddings(), le.get_output_embeddings())}.values()
                            if w.requires_grad]
            late = list({id(w): w for w in late + self._masked}.values())
        self.dp = GradBucketer(self.opt, bucket_bytes=bucket_bytes, process_group=process_group, late_params=late,
                               force_hooks=force_dp_hooks)
        if self.dp.active and self.dp.world > 1:
            # collectives share the CUs with backward: no persistent GEMM variant (ops.AVOID_PERSISTENT).  The flag is process-wide (the GEMM
            # call sites do not know their trainer): reference-counted, cleared when the LAST data-parallel trainer lets go (ADVICE r5:
            # restoring the value found at construction was only right for LIFO close order)
            ops.avoid_persistent_acquire()
            self.dp.on_remove = ops.avoid_persistent_release
        self.sched, self.base_lr, self.warmup, self.total = lr_scheduler, lr, warmup_steps, total_steps
        self.sched_step = 0
        self.overlap_optimizer, self._opt_stream, self._opt_event = bool(overlap_optimizer), None, None
        if self.overlap_optimizer:
            ve = getattr(model, "vision_encoder", None)
            if ve is None or any(p.requires_grad for p in ve.parameters()):
                raise ValueError("Trainer(overlap_optimizer=True) needs a frozen vision encoder (the forward that runs beside the update)")
            if shard is not None or graph:
                raise ValueError("Trainer(overlap_optimizer=True) is not combined with shard_optimizer (its step issues collectives) or graph")
        # the weight-gradient GEMMs add straight into the flat gradient buffer (functional.WGRAD_SINK); under data parallelism
        # the sink plays the post-accumulate hook for the bucketer
        self._sink = _WgradSink(self.opt, exclude=late, dp=self.dp if self.dp.active else None) if direct_wgrad else None

    def _opt_step(self, gscale):
        """clip + AdamW; with overlap_optimizer on a side stream behind everything queued so far (backward, the gradient exchange)"""
        lr = self.current_lr()
        if not (self.overlap_optimizer and self.opt.flat_g.is_cuda):
            self.opt.step(lr=lr, grad_scale=gscale)
            return
        if self._opt_stream is None:
            self._opt_stream = torch.cuda.Stream()
        main = torch.cuda.current_stream()
        self._opt_stream.wait_stream(main)
        with torch.cuda.stream(self._opt_stream):
            self.opt.step(lr=lr, grad_scale=gscale)
            ev = torch.cuda.Event()
            ev.record(self._opt_stream)
        self._opt_event = ev
        self.model._params_ready = ev          # consumed by Flamingo._params_barrier (before the first trainable parameter is read)

    def sync(self):
        """make the current stream wait for an optimizer update still running on the side stream (overlap_optimizer): call before reading
        parameters, gradients or optimizer buffers outside ``step()``"""
        if self._opt_event is not None:
            torch.cuda.current_stream().wait_event(self._opt_event)
            self._opt_event = None
            self.model._params_ready = None

    def close(self):
        """detach from the process: gradient hooks removed, ops.AVOID_PERSISTENT restored (a later single-rank Trainer in the same
        process gets the persistent GEMM variants back)"""
        self.dp.remove()

    def current_lr(self):
        if self.sched == "cosine":
            return cosine_lr(self.sched_step, self.base_lr, self.warmup, self.total)
        if self.sched == "linear":
            return linear_lr(self.sched_step, self.base_lr, self.warmup, self.total)
        return self.base_lr if self.sched_step >= self.warmup else self.base_lr * self.sched_step / max(1, self.warmup)

    def _fused_batch(self, batches):
        """GA micro-batches -> one batch: right-pad the token tensors to the longest, concatenate; ``weights`` carry the
        per-micro-batch loss normalisation (see __init__)."""
        L = max(b["lang_x"].shape[1] for b in batches)
        pad = self.ids["pad_id"]

        def padded(t, value):
            return t if t.shape[1] == L else torch.nn.functional.pad(t, (0, L - t.shape[1]), value=value)
        if len({tuple(b["vision_x"].shape[1:]) for b in batches}) != 1:
            raise ValueError("fuse_accum: the micro-batches of one optimizer step must carry the same number of images per sample")
        ids = torch.cat([padded(b["lang_x"], pad) for b in batches])
        labels, _ = ops.label_mask(ids, self.ids["answer_id"], self.ids["eoc_id"], self.ids["pad_id"], self.ids["media_id"], want_media_time=False)
        n_b = (labels[:, 1:] != -100).sum(1).float()                                   # labeled positions per sample
        sizes = tuple(b["lang_x"].shape[0] for b in batches)
        mb = self._mb_index.get(sizes)                    # sample -> micro-batch index; built once per size tuple (no per-step H2D / sync)
        if mb is None:
            mb = self._mb_index[sizes] = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes)).to(ids.device)
        n_mb = torch.zeros(len(sizes), device=ids.device).index_add_(0, mb, n_b).clamp_min(1.0)   # ... per micro-batch (small integers: exact)
        norm = (n_b.sum().clamp_min(1.0) / (len(batches) * n_mb))[mb]
        w = torch.cat([b["weights"].float() for b in batches])        # the task weights apply with and without --use_reweight (mmrec.py:203)
        return dict(vision_x=torch.cat([b["vision_x"] for b in batches]), lang_x=ids,
                    attention_mask=torch.cat([padded(b["attention_mask"], 0) for b in batches]), weights=w * norm)

    def _fuse_decide(self, batch):
        """automatic fuse_accum: take the fused pass only while its activations (GA x B x L tokens) stay inside FUSE_TOKEN_BUDGET.
        Decided ONCE, at the first micro-batch, and -- under data parallelism -- by ALL ranks together (MIN over the trainer's group:
        ranks that pad to different L near the budget would otherwise pick different paths).  Limitation, by design: later,
        longer batches are not re-checked (a path switch inside a run would change the StepOut.pending contract mid-epoch); a loader
        whose sequence lengths grow past the budget should pass fuse_accum=False or size UNIMP_FUSE_TOKENS for its longest batch."""
        B, L = batch["lang_x"].shape
        ok = self.grad_accum * B * L <= FUSE_TOKEN_BUDGET
        if self.dp.active and self.dp.world > 1:
            import torch.distributed as dist
            # a host tensor on gloo, a device tensor on RCCL (the backend's own device)
            dev = batch["lang_x"].device if dist.get_backend(self.dp.pg) == "nccl" else "cpu"
            t = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.dp.pg)
            ok = bool(int(t.item()))
        self.fuse_accum = ok
        self._fuse_auto = False

    def _fused_pass(self, batches, scale=1.0):
        loss, stats = self._micro_step(self._fused_batch(batches))
        self._mask_lm_head_grads()
        gscale = self.dp.finish() * scale
        self._opt_step(gscale)
        self.sched_step += 1
        self._last = (loss, stats)
        return StepOut(loss, stats)

    def _fused_step(self, batch):
        self._stash.append(batch)
        if len(self._stash) < self.grad_accum:
            if self._last is None:         # nothing computed yet: NaN loss (not a fake 0.0), stats of the right shape (focal_ce's [loss_sum, n_labeled, ce_sum])
                dev = batch["lang_x"].device
                self._last = (torch.full((), float("nan"), device=dev), torch.zeros(3, device=dev))
            return StepOut(self._last[0], self._last[1], pending=True)
        batches, self._stash = self._stash, []
        return self._fused_pass(batches)

    def flush(self, mean_over="ga"):
        """optimizer step over the micro-batches of an INCOMPLETE accumulation group (the loader ended inside it); a no-op returning None
        when nothing is pending.  accelerate does the same at the end of the dataloader (``sync_with_dataloader``: the last batch
        sets ``sync_gradients``), and it has divided every micro-batch loss by the full GA (``Accelerator.backward``), so the
        reference's partial group steps with sum(grad_i) / GA: that is ``mean_over="ga"``, the default.  ``mean_over="stashed"``
        averages over the k micro-batches actually seen (sum(grad_i) / k)."""
        if mean_over not in ("ga", "stashed"):
            raise ValueError(mean_over)
        if self.fuse_accum:
            k = len(self._stash)
            if k == 0:
                return None
            batches, self._stash = self._stash, []
            # _fused_batch normalises by the k micro-batches it is given (mean over k); the reference's partial group is k / GA of that
            return self._fused_pass(batches, scale=(k / self.grad_accum) if mean_over == "ga" else 1.0)
        k = self._micro % self.grad_accum if (self.grad_accum > 1 or self.use_graph) else 0
        if k == 0:
            return None
        self._mask_lm_head_grads()
        self.sync()
        gscale = self.dp.finish() / (self.grad_accum if mean_over == "ga" else k)
        self.dp.sync = True
        self._opt_step(gscale)
        self.sched_step += 1
        self._micro = 0
        return StepOut(*self._last) if self._last is not None else None

    def forward_loss(self, batch):
        ids = batch["lang_x"]
        labels, _ = ops.label_mask(ids, self.ids["answer_id"], self.ids["eoc_id"], self.ids["pad_id"], self.ids["media_id"],
                                   want_media_time=False)
        vx = batch["vision_x"].unsqueeze(2) if batch["vision_x"].ndim == 5 else batch["vision_x"]
        if self.sparse_head:
            bj = (labels[:, 1:] != -100).nonzero()                       # (b, j): position j predicts the labeled token j+1
            if bj.shape[0] > 0:
                rows = bj[:, 0] * ids.shape[1] + bj[:, 1]
                out = self.model(vision_x=vx, lang_x=ids, attention_mask=batch["attention_mask"], labels=None, head_rows=rows)
                w = self.model.lang_encoder.get_output_embeddings().weight
                loss, stats = F_.sparse_head_loss(out.hidden_rows, w, labels[bj[:, 0], bj[:, 1] + 1].contiguous(),
                                                  batch["weights"].float()[bj[:, 0]].contiguous(), self.gamma, self.use_reweight)
                return loss, stats, out, labels
        if self.dense_head_backward:
            out = self.model(vision_x=vx, lang_x=ids, attention_mask=batch["attention_mask"], labels=None)
            loss, stats = F_.focal_ce(out["logits"], labels, batch["weights"], self.gamma, self.use_reweight)
            return loss, stats, out, labels
        pending = self._labeled_rows_begin(labels)                        # queued BEFORE the forward, read after it is queued
        out = self.model(vision_x=vx, lang_x=ids, attention_mask=batch["attention_mask"], labels=None, head_rows="hidden")
        rows = self._labeled_rows_end(pending)                            # host wait: only for the count, long done by now
        w = self.model.lang_encoder.get_output_embeddings().weight
        loss, stats, logits = F_.dense_head_loss(out.hidden_rows, w, labels, batch["weights"], rows, self.gamma, self.use_reweight)
        out.logits, out.hidden_rows = logits, None                       # the reference's dense output["logits"] (mmrec.py:190)
        return loss, stats, out, labels

    def _labeled_rows_begin(self, labels):
        """The compact head backward needs the scored positions as flat row indices b * L + j (position j predicts the labeled token
        j + 1) and their COUNT on the host.  ``nonzero()`` would stop the host right here, at the top of the step, until the device has
        drained the previous step -- and the device then idles while the host queues the first kernels of this one (6 % of the
        step at the reference's b = 3 x GA 2).  Instead the indices are formed at a fixed size (stable sort of the mask: scored
        positions first, ascending), the count goes to pinned memory behind an event, and the host reads it only after the
        model forward has been queued."""
        B, L = labels.shape
        m = torch.zeros(B, L, dtype=torch.uint8, device=labels.device)
        m[:, :-1] = labels[:, 1:] != -100
        flat = m.view(-1)
        if not labels.is_cuda or _ROWS_SYNC:
            return flat.nonzero().view(-1), None
        order = torch.sort(flat, descending=True, stable=True).indices
        if getattr(self, "_cnt_host", None) is None:
            self._cnt_host = torch.empty(1, dtype=torch.int64, pin_memory=True)
        self._cnt_host.copy_(flat.sum(dtype=torch.int64).view(1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return order, ev

    def _labeled_rows_end(self, pending):
        order, ev = pending
        if ev is None:
            return order
        ev.synchronize()
        return order[:int(self._cnt_host[0])]

    def _mask_lm_head_grads(self):
        if not self._masked:
            return
        self.opt._reattach()          # a stray .grad (foreign code) is folded into the flat view FIRST: the mask must see the whole gradient
        a = self.ids["answer_id"]
        for w in self._masked:
            g = w.grad
            keep = g[a].clone()
            g.zero_()
            g[a].copy_(keep)

    def _backward(self, loss):
        F_.WGRAD_SINK = self._sink
        if self._sink is not None:
            self._sink.seen.clear()
        try:
            loss.backward()
        finally:
            F_.WGRAD_SINK = None

    def _micro_step(self, batch):
        """forward + loss + backward of one micro-batch; gradients ADD into the flat buffer (zeroed by the optimizer kernel only)."""
        loss, stats, out, _ = self.forward_loss(batch)
        self._backward(loss)
        return loss.detach(), stats

    _GRAPH_KEYS = ("vision_x", "lang_x", "attention_mask", "weights")

    def _graphed_micro_step(self, batch):
        key = tuple((k, tuple(batch[k].shape), batch[k].dtype) for k in self._GRAPH_KEYS)
        g = self._graph
        if g is None or g["key"] != key:
            if g is None or g.get("warm_key") != key:          # first micro-step of this shape: eager
                self._graph = {"key": None, "warm_key": key}
                return self._micro_step(batch)
            static = {k: batch[k].clone() for k in self._GRAPH_KEYS}
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph):                      # recorded, not executed: the replay below runs this micro-batch
                loss, stats = self._micro_step(static)
            g = self._graph = {"key": key, "graph": graph, "static": static, "loss": loss, "stats": stats}
        else:
            for k in self._GRAPH_KEYS:
                g["static"][k].copy_(batch[k])
        g["graph"].replay()
        return g["loss"], g["stats"]

    def step(self, batch):
        """returns (loss, stats) device tensors.  The default loss path takes the number of labeled positions on the host: ONE host
        wait per micro-step, for a count that is queued before the model forward and read after it (_labeled_rows_begin: the
        device does not idle); ``sparse_head`` needs it before the forward; ``dense_head_backward=True`` (and ``graph=True``, which
        implies it) has none."""
        self.model.train()
        if self._fuse_auto:
            self._fuse_decide(batch)
        if self.fuse_accum:
            return self._fused_step(batch)
        micro = self._graphed_micro_step if self.use_graph else self._micro_step
        if self.grad_accum > 1 or self.use_graph:
            # micro-batches add their gradients into the flat buffer (zeroed by the optimizer kernel only); ranks exchange
            # once, after the last one; 1/GA is folded into the optimizer's gradient scale like 1/W.  The LR schedule
            # advances once per optimizer step (mmrec.py:691-692 sizes its schedule in optimizer steps).
            self._micro += 1
            self.dp.sync = False          # hooks only fold gradients; finish() issues the exchange (a replayed graph runs no hook)
            loss, stats = micro(batch)
            self._last = (loss, stats)
            if self._micro % self.grad_accum == 0:
                self._mask_lm_head_grads()
                gscale = self.dp.finish() / self.grad_accum
                self.dp.sync = True
                self._opt_step(gscale)
                self.sched_step += 1
            return StepOut(loss, stats)
        loss, stats = micro(batch)
        self._mask_lm_head_grads()
        gscale = self.dp.finish()
        self._opt_step(gscale)
        self.sched_step += 1
        return StepOut(loss, stats)
