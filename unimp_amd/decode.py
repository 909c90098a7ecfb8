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
    def prefill(self, input_ids):
        m = self.model
        R, L0 = input_ids.shape
        K = self.beams
        self.L0 = L0
        prompts = input_ids[::K].contiguous() if K > 1 else input_ids          # one row per beam group
        n_layers = len(self.lm._get_decoder_layers())
        pc = F_.DecodeCache(n_layers, capacity=L0 if K > 1 else L0 + self.max_new)
        m._condition_media_locations(input_ids=prompts)
        out = self.lm(input_ids=prompts, past_key_values=pc, use_cache=True, logits_last_only=True)
        logits = out["logits"][:, -1]
        if K > 1:                               # give every beam its copy of the prompt's K/V, projected media and vision rows
            c = F_.DecodeCache(n_layers, capacity=L0 + self.max_new)
            c._reserve(pc.kv.new_empty((R, 0) + tuple(pc.kv.shape[4:])), L0)     # [R, 0, nh, hd] template: allocates the full tensor
            c.kv[:, :, :, :L0].copy_(pc.kv[:, :, :, :L0].repeat_interleave(K, 2))
            B = prompts.shape[0]
            for lk, lp in zip(c.layers, pc.layers):          # xkv is [B * T*n, 2*inner]: repeat whole prompts, not single key rows
                lk.xkv = (lp.xkv.view(B, lp.xkv.shape[0] // B, -1).repeat_interleave(K, 0).reshape(-1, lp.xkv.shape[1])
                          if lp.xkv is not None else None)
            c.len = L0
            m._repeat_conditioned_vision(K)
            logits = logits.repeat_interleave(K, 0)
        else:
            c = pc
        self.cache = c
        c.shared_prefix = L0 if self.reorder else 0
        c.media_count = (input_ids == m.media_token_id).sum(1, keepdim=True).to(torch.int32)
        dev = input_ids.device
        self.tok = torch.zeros((R, 1), dtype=torch.long, device=dev)
        self.src = torch.arange(R, dtype=torch.long, device=dev)
        return logits

    # -- one token per row
    def _step_body(self):
        c = self.cache
        if self.reorder:                       # generated tail only: the prompt part is identical in all beams of an item
            tail = c.kv[:, :, :, self.L0:self.L0 + self.max_new]
            tail.copy_(tail.index_select(2, self.src))
        self.model._condition_cached_media(c, 1)
        out = self.lm(input_ids=self.tok, past_key_values=c, use_cache=True)
        c.step.advance()
        return out["logits"][:, -1]

    def step(self, tokens, src=None):
        """tokens [rows] = the token each row was extended with; src [rows] = the row whose hypothesis it continues."""
        c = self.cache
        if c.len >= self.L0 + self.max_new:
            raise RuntimeError("DecodeSession: more steps than max_new_tokens")
        if c.step is None:
            c.step = F_.StepState(self.tok.shape[0], c.len, self.tok.device)
        self.tok.copy_(tokens.view(-1, 1))
        if src is not None:
            self.src.copy_(src)
        if not self.use_graph:
            logits = self._step_body()
        elif self.steps == 0:
            logits = self._step_body()          # first step eagerly: warms every lazy initialisation outside the capture
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
