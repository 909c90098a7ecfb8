"""Oracle: the host half of one optimizer step (fp32, CPU).  TEST INFRASTRUCTURE ONLY.

Restates UniMP/mmrec.py:train_one_epoch piece by piece:
  label_mask_loop / label_mask     mmrec.py:143-168
  weighted_focal_ce                mmrec.py:190-213
  focal_ce_dlogits (analytic)      derivative of the above (focal term NOT detached)
  grouped_params                   mmrec.py:609-631
  cosine_lr                        mmrec.py:687-693 (transformers get_cosine_schedule_with_warmup)
  clip_grad_norm_ / adamw_step     mmrec.py:247-256, 671 (torch.optim.AdamW defaults)
PINNED against the reference itself: tests/golden/train_step_*.npz were captured from
``mmrec.train_one_epoch`` run in the build container (oracle/make_golden.py).
"""
import math
import numpy as np
import torch
import torch.nn.functional as F


def label_mask_loop(input_ids, answer_id, eoc_id, pad_id, media_id):
    """Literal restatement of the reference's state machine (mmrec.py:143-168). numpy int64."""
    labels = np.array(input_ids, dtype=np.int64, copy=True)
    for i in range(labels.shape[0]):
        flag = 0
        for j in range(labels.shape[1]):
            if not flag:
                if labels[i, j] == answer_id:
                    flag = 1
                labels[i, j] = -100
            else:
                if labels[i, j] == eoc_id:
                    flag = 0
                    labels[i, j] = -100
    labels[labels == pad_id] = -100
    labels[:, 0] = -100
    labels[labels == answer_id] = -100
    labels[labels == media_id] = -100
    return labels


def label_mask(input_ids, answer_id, eoc_id, pad_id, media_id):
    """Closed form: keep[j] <=> the last {<answer>,<eoc>} event strictly before j
    is an <answer>, and tok not in {eoc, answer, pad, image}, and j > 0."""
    ids = np.asarray(input_ids, dtype=np.int64)
    L = ids.shape[1]
    idx = np.arange(L)[None]
    last_ans = np.maximum.accumulate(np.where(ids == answer_id, idx, -1), 1)
    last_eoc = np.maximum.accumulate(np.where(ids == eoc_id, idx, -1), 1)
    # events strictly before j: shift right by one
    la = np.concatenate([np.full((ids.shape[0], 1), -1), last_ans[:, :-1]], 1)
    le = np.concatenate([np.full((ids.shape[0], 1), -1), last_eoc[:, :-1]], 1)
    keep = (la > le) & (ids != eoc_id) & (ids != answer_id) & (ids != pad_id) & (ids != media_id) & (idx > 0)
    return np.where(keep, ids, -100)


def weighted_focal_ce(logits, labels, weights, gamma, use_reweight=True):
    """mmrec.py:190-213.  logits (B,L,V) float; labels (B,L) with -100; weights (B,)."""
    n1, n2 = labels.shape[0], labels.shape[1] - 1
    shift_logits = logits[:, :-1, :].contiguous().view(-1, logits.size(-1))
    lab = labels[:, 1:].contiguous().view(-1)
    lm_loss = F.cross_entropy(shift_logits, lab, reduction="none").view(n1, n2)
    loss = (weights.unsqueeze(1) * lm_loss).view(-1)
    if use_reweight:
        p = F.softmax(shift_logits, dim=-1)
        pt = p[torch.arange(len(shift_logits)), lab]      # label -100 wraps to column V-100 (quirk)
        loss = loss * (1 - pt) ** gamma
    return loss.sum() / (lab != -100).sum()


def focal_ce_dlogits(logits, labels, weights, gamma, use_reweight=True):
    """Analytic d loss / d logits (B,L,V) of weighted_focal_ce."""
    B, L, V = logits.shape
    z = logits[:, :-1].double()
    lab = labels[:, 1:]
    valid = lab != -100
    n = valid.sum().double()
    p = z.softmax(-1)
    y = lab.clamp(min=0)
    py = p.gather(-1, y[..., None]).squeeze(-1)
    if use_reweight:
        coef = (1 - py) ** gamma - gamma * py * (1 - py) ** (gamma - 1) * py.log() if gamma != 0 \
            else torch.ones_like(py)
    else:
        coef = torch.ones_like(py)
    g = p.clone()
    g.scatter_add_(-1, y[..., None], -torch.ones_like(py)[..., None])
    g = g * (coef * weights.double()[:, None] / n * valid)[..., None]
    out = torch.zeros(B, L, V, dtype=torch.float64)
    out[:, :-1] = g
    return out


def grouped_params(named_parameters, weight_decay):
    """mmrec.py:609-631.  NOTE the quirk: ``ff.0.weight`` (a LayerNorm gain) decays."""
    def apply_decay(x):
        return ("gated_cross_attn_layer" in x and "ff_gate" not in x and "attn_gate" not in x
                and "norm" not in x and "bias" not in x)
    wd, nowd = [], []
    for n, p in named_parameters:
        (wd if apply_decay(n) else nowd).append((n, p))
    return [{"params": wd, "weight_decay": weight_decay}, {"params": nowd, "weight_decay": 0.0}]


def cosine_lr(step, base_lr, warmup_steps, total_steps, num_cycles=0.5):
    """transformers.get_cosine_schedule_with_warmup lambda (used at mmrec.py:687-693)."""
    if step < warmup_steps:
        return base_lr * float(step) / float(max(1, warmup_steps))
    prog = float(step - warmup_steps) / float(max(1, total_steps - warmup_steps))
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * num_cycles * 2.0 * prog)))


def clip_coef(grads, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_: total L2 norm, coef = min(1, max_norm/(norm+1e-6))."""
    tot = torch.sqrt(sum((g.double() ** 2).sum() for g in grads))
    return float(tot), float(min(1.0, max_norm / (float(tot) + 1e-6)))


def adamw_step(p, g, m, v, step, lr, wd, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.AdamW single-tensor update (step is 1-based), fp32 in place."""
    p.mul_(1 - lr * wd)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
