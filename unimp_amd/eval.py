"""Recommendation evaluation around ``Flamingo.generate`` (UniMP/pipeline/eval/eval_rec.py:32-190, metrics from
UniMP/pipeline/eval/rec_metrics.py:26-37,62-100,106-111): K = 10 beams, 10 returned sequences, 50 new tokens per user; a hit
is a returned hypothesis whose text after the last "?" equals the target item string; HR / NDCG / MRR at 3, 5, 10."""
import numpy as np
import torch


def mrr_at_k(r, k):
    nz = np.asarray(r)[:k].nonzero()[0]
    return 1.0 / (nz[0] + 1) if len(nz) else 0.0


def hit_at_k(r, k):
    return 1.0 if np.sum(np.array(r)[:k]) > 0 else 0.0


def dcg_at_k(r, k, method=1):
    r = np.asarray(r, dtype=float)[:k]
    if not r.size:
        return 0.0
    if method == 0:
        return r[0] + np.sum(r[1:] / np.log2(np.arange(2, r.size + 1)))
    if method == 1:
        return np.sum(r / np.log2(np.arange(2, r.size + 2)))
    raise ValueError("method must be 0 or 1.")


def ndcg_at_k(r, k, len_gt, method=1):
    ideal = [1.0] * k if len_gt > k else [1.0] * len_gt + [0.0] * (k - len_gt)
    dcg_max = dcg_at_k(ideal, k, method)
    return dcg_at_k(r, k, method) / dcg_max if dcg_max else 0.0


def relevance(texts, target, K=10):
    """eval_rec.py:111-127: decoded hypotheses -> binary relevance vector of length 10 (rank order = beam order)."""
    texts = [t.split("</s>")[0] for t in texts]
    texts = [t.split("?")[-1].strip() for t in texts]
    hits = np.array([t == target for t in texts], dtype=int)
    r = np.array([0] * 10)
    r[:len(hits)] = hits
    return r


def user_metrics(r, K=10):
    out = {}
    for k in (3, 5, K):
        out[f"hr@{k}"], out[f"ndcg@{k}"], out[f"mrr@{k}"] = hit_at_k(r, k), ndcg_at_k(r, k, 1), mrr_at_k(r, k)
    return out


@torch.no_grad()
def eval_model_rec(model, samples, tokenizer, K=10, max_new_tokens=50, image_preprocessor=None, device="cuda", users_per_batch=1):
    """samples: iterable of eval samples (``RecDataset(split != "train")[i]``, one user each, as UniMP's eval loader yields).
    ``image_preprocessor``: an ``ImagePreprocessor`` for samples that carry raw uint8 images (``defer_images=True``).
    ``users_per_batch`` > 1 decodes several users per ``generate`` call (right-padded prompts, per-row positions): the
    decode step is bound by the weight reads, which the users then share.  Returns the mean of every metric over the users."""
    model.eval()
    rows, group = [], []

    def flush():
        if not group:
            return
        imgs = []
        for s in group:
            im = s["net_input"]["patch_images"]
            if isinstance(im, (list, tuple)):
                if image_preprocessor is None:
                    raise ValueError("eval_model_rec: raw images need an image_preprocessor")
                im = image_preprocessor(im)
            imgs.append(im.to(device=device, dtype=torch.bfloat16))
        vision_x = torch.stack(imgs).unsqueeze(2)                                        # (users, T, 1, 3, H, W)
        lens = [s["net_input"]["input_ids"].numel() for s in group]
        L = max(lens)
        pad = tokenizer.pad_token_id if tokenizer.pad_token_id is not None else tokenizer.eos_token_id
        ids = torch.full((len(group), L), pad, dtype=torch.long)
        for i, s in enumerate(group):
            ids[i, :lens[i]] = s["net_input"]["input_ids"]
        mask = (torch.arange(L)[None, :] < torch.tensor(lens)[:, None]).long()
        gen = model.generate(vision_x=vision_x, lang_x=ids.to(device), attention_mask=mask.to(device), num_beams=K,
                             num_return_sequences=K, early_stopping=True, max_new_tokens=max_new_tokens,
                             eos_token_id=tokenizer.eos_token_id, pad_token_id=tokenizer.eos_token_id)
        texts = tokenizer.batch_decode(gen, skip_special_tokens=True)
        for i, s in enumerate(group):
            rows.append(user_metrics(relevance(texts[i * K:(i + 1) * K], s["net_output"]["output_ids"], K), K))
        group.clear()

    for s in samples:
        group.append(s)
        if len(group) == users_per_batch:
            flush()
    flush()
    return {k: float(np.mean([r[k] for r in rows])) for k in rows[0]} if rows else {}
