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
    def __init__(self, model, max_new_tokens, reorder, graph=True):
        """model: Flamingo with the vision features already conditioned (``_encode_vision_x``).  ``reorder``: beam search
        (rows exchange hypotheses between steps); ``graph=False`` keeps launching the step kernel by kernel."""
        self.model, self.lm = model, model.lang_encoder
        self.max_new, self.reorder, self.use_graph = max_new_tokens, reorder, graph
        self.cache = None
        self.graph = None
        self.steps = 0

    # -- prompt
    def prefill(self, input_ids):
        m = self.model
        R, L0 = input_ids.shape
        self.L0 = L0
        self.cache = F_.DecodeCache(len(self.lm._get_decoder_layers()), capacity=L0 + self.max_new)
        self.cache.shared_prefix = L0 if self.reorder else 0
        m._condition_media_locations(input_ids=input_ids)
        out = self.lm(input_ids=input_ids, past_key_values=self.cache, use_cache=True, logits_last_only=True)
        self.cache.media_count = (input_ids == m.media_token_id).sum(1, keepdim=True).to(torch.int32)
        dev = input_ids.device
        self.tok = torch.zeros((R, 1), dtype=torch.long, device=dev)
        self.src = torch.arange(R, dtype=torch.long, device=dev)
        return out["logits"][:, -1]

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
