"""KV-cached decode of a media-conditioned Flamingo (call sites UniMP/pipeline/eval/eval_rec.py:100-110:
K = 10 beams, 50 new tokens per user).

One prefill over the prompt fills the cache; after that a step feeds one token per row.  A step is launch-bound (about
600 small kernels: 48 decoder / cross-attention blocks on a handful of rows), so from the second step on it is replayed
as ONE HIP graph: the position lives on the device (functional.StepState), every buffer the step touches is static, and
the beam reorder of the generated K/V tail is part of the graph.  The host only writes the new tokens / source rows and
reads the logits.
"""
import torch

from . import functional as F_
from . import ops


# step signatures that have run eagerly once in this process: everything lazy behind them (live GEMM tuning, kernel attributes, the decode attention's
# workspace) exists, so a later session of the same signature captures its FIRST step instead of running it eagerly (8 - 10 ms per generate() call)
_WARM = set()


class DecodeSession:
    def __init__(self, model, max_new_tokens, reorder, graph=True, beams=1):
        """model: Flamingo with the vision features of the PROMPTS already conditioned (``_encode_vision_x``, one row per
        prompt).  ``reorder``: beam search (rows exchange hypotheses between steps); ``graph=False`` keeps launching the
        step kernel by kernel; ``beams``: rows per prompt -- the prompt is prefilled ONCE per prompt and its K/V, projected
        media and vision features are then repeated for the beams (the rows of a beam group are identical at that point)."""
        self.model, self.lm = model, model.lang_encoder
        self.max_new, self.reorder, self.use_graph, self.beams = max_new_tokens, reorder, graph, beams
        self.cache = None
        self.graph = None
        self.steps = 0

    # -- prompt
    def prefill(self, prompts, lengths=None):
        """prompts [B, Lmax] (right-padded when ``lengths`` [B] is given) -> logits [B * beams, V] of each prompt's last token."""
        m = self.model
        B, Lmax = prompts.shape
        K = self.beams
        R = B * K
        dev = prompts.device
        self.host_len = [Lmax] * B if lengths is None else [int(x) for x in lengths.tolist()]
        n_layers = len(self.lm._get_decoder_layers())
        pc = F_.DecodeCache(n_layers, capacity=Lmax if K > 1 else Lmax + self.max_new)
        m._condition_media_locations(input_ids=prompts)
        if lengths is None:
            out = self.lm(input_ids=prompts, past_key_values=pc, use_cache=True, logits_last_only=True)
            row_len = torch.full((B,), Lmax, dtype=torch.long, device=dev)
        else:
            mask = (torch.arange(Lmax, device=dev)[None, :] < lengths.to(dev)[:, None]).long()
            out = self.lm(input_ids=prompts, attention_mask=mask, past_key_values=pc, use_cache=True,
                          last_index=lengths.to(dev).long() - 1)
            row_len = lengths.to(dev).long()
        logits = out["logits"][:, -1]
        if K > 1:                               # give every beam its copy of the prompt's K/V, projected media and vision rows
            c = F_.DecodeCache(n_layers, capacity=Lmax + self.max_new)
            c._reserve(pc.kv.new_empty((R, 0) + tuple(pc.kv.shape[4:])), Lmax)    # [R, 0, nh, hd] template: allocates the full tensor
            c.kv[:, :, :, :Lmax].copy_(pc.kv[:, :, :, :Lmax].repeat_interleave(K, 2))
            for lk, lp in zip(c.layers, pc.layers):          # xkv is [B * T*n, 2*inner]: repeat whole prompts, not single key rows
                lk.xkv = (lp.xkv.view(B, lp.xkv.shape[0] // B, -1).repeat_interleave(K, 0).reshape(-1, lp.xkv.shape[1])
                          if lp.xkv is not None else None)
            m._repeat_conditioned_vision(K)
            logits = logits.repeat_interleave(K, 0)
            row_len = row_len.repeat_interleave(K)
        else:
            c = pc
        c.len = Lmax                             # capacity bookkeeping on the host; the rows' true lengths live in StepState
        c.row_len = None
        self.cache, self.Lmax = c, Lmax
        c.media_count = ((prompts == m.media_token_id) & (torch.arange(Lmax, device=dev)[None, :] < row_len[::K, None])) \
            .sum(1, keepdim=True).to(torch.int32).repeat_interleave(K, 0)
        c.step = F_.StepState(R, row_len, dev)
        if K > 1:              # the K rows of a prompt share its prefix: the decode attention reads those keys once per prompt
            c.group, c.shared_len = K, row_len[::K].to(torch.int32).contiguous()
        self.tok = torch.zeros((R, 1), dtype=torch.long, device=dev)
        self.src = torch.arange(R, dtype=torch.long, device=dev)
        self.base = (torch.arange(R, device=dev) // K) * K               # first row of each row's beam group
        self.slot0 = torch.tensor(self.host_len, dtype=torch.int32, device=dev)      # first slot of every group's generated tail
        return logits

    # -- one token per row
    def _step_body(self):
        c = self.cache
        if self.reorder:          # generated tail only (a group's rows share their prompt), group by group: tails start at the prompt lengths
            local = self.src - self.base
            if F_.DECODE_FUSED and self.beams <= 16 and c.kv.is_contiguous() and (c.kv.shape[4] * c.kv.shape[5]) % 8 == 0:
                # one launch, one pass, only the slots generated so far (csrc/elementwise.hip kv_reorder_beams_kernel)
                ops.kv_reorder_beams(c.kv, self.beams, local, self.slot0, c.step.pos_idx, self.max_new)
            else:
                for b, L0 in enumerate(self.host_len):
                    tail = c.kv[:, :, b * self.beams:(b + 1) * self.beams, L0:L0 + self.max_new]
                    tail.copy_(tail.index_select(2, local[b * self.beams:(b + 1) * self.beams]))
        self.model._condition_cached_media(c, 1)
        out = self.lm(input_ids=self.tok, past_key_values=c, use_cache=True)
        c.step.advance()
        return out["logits"][:, -1]

    def step(self, tokens, src=None):
        """tokens [rows] = the token each row was extended with; src [rows] = the row whose hypothesis it continues (a row
        of the SAME beam group: hypotheses never move between prompts; not checked -- that would need a host sync)."""
        c = self.cache
        if self.steps >= self.max_new:
            raise RuntimeError("DecodeSession: more steps than max_new_tokens")
        self.tok.copy_(tokens.view(-1, 1))
        if src is not None:
            self.src.copy_(src)
        # what the lazily created state depends on: the model, the rows, the head geometry and the number of 128-key chunks of the cache (the decode
        # attention's workspace) -- not the exact capacity, so prompts of other lengths in the same chunk count are warm as well
        kvs = c.kv.shape
        sig = (id(self.model), kvs[0], kvs[2], (kvs[3] + 127) // 128, kvs[4], kvs[5], self.beams, bool(self.reorder), F_.DECODE_FUSED, F_.DECODE_STEP_ATTN,
               F_.DECODE_STEP_GROUPED)
        if not self.use_graph:
            logits = self._step_body()
        elif self.graph is None and sig not in _WARM:
            logits = self._step_body()          # first step of this signature eagerly: warms every lazy initialisation outside the capture
            _WARM.add(sig)
        else:
            if self.graph is None:
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    self._logits = self._step_body()
            self.graph.replay()
            logits = self._logits
        self.steps += 1
        c.len += 1
        return logits
