"""Decoding loops behind ``Flamingo.generate`` (call sites UniMP/pipeline/eval/eval_rec.py:100-110:
``num_beams=10, num_return_sequences=10, early_stopping=True, max_new_tokens=50, eos_token_id=pad_token_id=eos``).

Host logic only: the scores come from ``logits_fn(sequences) -> [rows, V]`` (last-position logits), which the model
implements on the HIP kernels.  Semantics follow transformers' ``generate`` for a decoder-only model -- greedy search and
beam search with the BeamSearchScorer rules (2K candidates per step, EOS candidates only count inside the top K ranks,
finished hypotheses scored ``sum_logprobs / generated_len ** length_penalty``, ``early_stopping=True`` stops a batch item
once K hypotheses are finished, remaining beams are finalised at ``max_new_tokens``).  Pinned against the installed
transformers on a tiny GPT-NeoX in tests/test_generate_cpu.py.  No KV cache yet: every step re-scores the whole sequence.
"""
import os

import torch

# beam search: log_softmax + beam scores + top-2K of a step in two launches of the HIP library (GPU logits only; 0: the torch ops)
BEAM_TOPK_FUSED = os.environ.get("UNIMP_BEAM_TOPK_FUSED", "1") != "0"


def _fused_topk(logits, beam_scores, K):
    """(scores, flat indices) of the step's 2K candidates per prompt through ops.beam_topk, or None where it does not apply (CPU logits, shapes)"""
    if not (BEAM_TOPK_FUSED and logits.is_cuda):
        return None
    from . import ops
    if not ops.beam_topk_ok(logits, K, 2 * K):
        return None
    return ops.beam_topk(logits, beam_scores, K, 2 * K)


def _ban_repeated_ngrams(seqs, logp, n):
    if n <= 0 or seqs.shape[1] + 1 < n:
        return
    for r in range(seqs.shape[0]):
        toks = seqs[r].tolist()
        prefix = tuple(toks[len(toks) - n + 1:]) if n > 1 else ()
        banned = [toks[i + n - 1] for i in range(len(toks) - n + 1) if tuple(toks[i:i + n - 1]) == prefix]
        if banned:
            logp[r, banned] = -float("inf")


@torch.no_grad()
def greedy_search(logits_fn, input_ids, max_new_tokens, eos_token_id=None, pad_token_id=None):
    seqs = input_ids.clone()
    B = seqs.shape[0]
    unfinished = torch.ones(B, dtype=torch.bool, device=seqs.device)
    pad = pad_token_id if pad_token_id is not None else (eos_token_id if eos_token_id is not None else 0)
    for _ in range(max_new_tokens):
        nxt = logits_fn(seqs).float().argmax(-1)
        nxt = torch.where(unfinished, nxt, torch.full_like(nxt, pad))
        seqs = torch.cat([seqs, nxt[:, None]], 1)
        if eos_token_id is not None:
            unfinished = unfinished & (nxt != eos_token_id)
            if not unfinished.any():
                break
    return seqs


class _Hyps:
    def __init__(self, k, length_penalty, early_stopping):
        self.k, self.lp, self.early, self.beams, self.worst = k, length_penalty, early_stopping, [], 1e9

    def add(self, tokens, sum_logprobs, generated_len):
        score = sum_logprobs / (generated_len ** self.lp)
        if len(self.beams) < self.k or score > self.worst:
            self.beams.append((score, tokens))
            if len(self.beams) > self.k:
                self.beams.sort(key=lambda t: t[0])
                del self.beams[0]
                self.worst = self.beams[0][0]
            else:
                self.worst = min(score, self.worst)

    def is_done(self, best_sum_logprobs, generated_len):
        if len(self.beams) < self.k:
            return False
        if self.early is True:
            return True
        return self.worst >= best_sum_logprobs / (generated_len ** self.lp)


@torch.no_grad()
def beam_search(logits_fn, input_ids, num_beams, max_new_tokens, eos_token_id, pad_token_id=None, num_return_sequences=1,
                early_stopping=True, length_penalty=1.0, no_repeat_ngram_size=0, stateful=False, trace=None):
    """input_ids [B, L0] -> [B * num_return_sequences, <= L0 + max_new_tokens] (right-padded with pad_token_id).
    trace: optional list; every step appends (scores, flat indices beam * V + token) of the sorted top-2K candidates per
    batch item (tests measure the margins that decide which beams survive)."""
    dev = input_ids.device
    B, L0 = input_ids.shape
    K = num_beams
    pad = pad_token_id if pad_token_id is not None else eos_token_id
    seqs = input_ids.repeat_interleave(K, 0)
    beam_scores = torch.zeros(B, K, dtype=torch.float32, device=dev)
    beam_scores[:, 1:] = -1e9
    hyps = [_Hyps(K, length_penalty, early_stopping) for _ in range(B)]
    done = [False] * B
    src = None
    for step in range(max_new_tokens):
        cur_len = seqs.shape[1]
        logits = logits_fn(seqs, src) if stateful else logits_fn(seqs)
        V = logits.shape[-1]
        fused = _fused_topk(logits, beam_scores, K) if no_repeat_ngram_size <= 0 else None
        if fused is not None:                      # log_softmax + beam scores + top-2K in two launches (csrc/elementwise.hip beam_topk_a / _b)
            top_s, top_i = fused
        else:
            logp = torch.log_softmax(logits.float(), -1)
            _ban_repeated_ngrams(seqs, logp, no_repeat_ngram_size)
            scores = (logp + beam_scores.view(-1, 1)).view(B, K * V)
            top_s, top_i = torch.topk(scores, 2 * K, dim=1, largest=True, sorted=True)
        top_s_h, top_i_h = top_s.tolist(), top_i.tolist()
        if trace is not None:
            trace.append((top_s_h, top_i_h))
        new_scores = torch.zeros(B, K, dtype=torch.float32)
        new_tok = torch.full((B, K), pad, dtype=torch.long)
        new_src = torch.zeros(B, K, dtype=torch.long)
        for b in range(B):
            if done[b]:
                new_src[b] = b * K
                continue
            n = 0
            for rank in range(2 * K):
                s, idx = top_s_h[b][rank], top_i_h[b][rank]
                beam, tok = idx // V, idx % V
                if eos_token_id is not None and tok == eos_token_id:
                    if rank >= K:
                        continue
                    hyps[b].add(seqs[b * K + beam].clone(), s, cur_len + 1 - L0)
                else:
                    new_scores[b, n], new_tok[b, n], new_src[b, n] = s, tok, b * K + beam
                    n += 1
                if n == K:
                    break
            done[b] = done[b] or hyps[b].is_done(max(top_s_h[b]), cur_len + 1 - L0)
        src = new_src.view(-1).to(dev)
        seqs = torch.cat([seqs[src], new_tok.view(-1, 1).to(dev)], 1)
        beam_scores = new_scores.to(dev)
        if all(done):
            break
    cur_len = seqs.shape[1]
    bs = beam_scores.tolist()
    for b in range(B):
        if not done[b]:
            for k in range(K):
                hyps[b].add(seqs[b * K + k].clone(), bs[b][k], cur_len - L0)
    R = num_return_sequences
    best = []
    for b in range(B):
        ranked = sorted(hyps[b].beams, key=lambda t: t[0], reverse=True)[:R]
        best += [t[1] for t in ranked]
    max_len = min(max(len(t) for t in best) + 1, L0 + max_new_tokens)
    out = torch.full((len(best), max_len), pad, dtype=torch.long, device=dev)
    for i, t in enumerate(best):
        out[i, :len(t)] = t
        if len(t) < max_len and eos_token_id is not None:
            out[i, len(t)] = eos_token_id
    return out
