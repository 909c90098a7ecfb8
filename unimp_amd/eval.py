"""The reference's five evaluation loops around ``Flamingo.generate`` (UniMP/pipeline/eval/), one user per sample as its eval
loader yields them; every loop can decode several users per ``generate`` call (``users_per_batch``: right-padded prompts share
the weight reads of the decode step).

  eval_model_rec      eval_rec.py:32-190      K = 10 beams, 10 returned, 50 new tokens; hit = text after the last "?" equals the
                                              target item string; HR / NDCG / MRR @ 3, 5, 10 (rec_metrics.py:26-37,62-100,106-111)
  eval_model_search   eval_search.py:29-175   the same ranking metrics, 20 new tokens
  eval_model_exp      eval_exp.py:31-205      5 beams, 1 returned, 256 new tokens; "rate_R explanation ..." -> MAE / RMSE of the
                                              rating (3.0 when it does not parse) + BLEU-1 precision, ROUGE-1/2/L of the text
  eval_model_img_sel  eval_img_sel.py:29-148  2 beams, 1 returned, 40 new tokens; set of generated "s_i" tokens vs the right ones:
                                              recall / precision / F1
  eval_model_img_gen  eval_img_gen.py:29-156  greedy, 600 new tokens (the image-token sequence of the next item); returns
                                              {item: [text]} -- the reference only dumps it to JSON, its metrics are commented out

The text metrics of eval_exp come from the third-party ``evaluate`` hub (bleu / rouge / meteor; not installed, not under
/root/reference): BLEU's unigram modified precision and ROUGE F-measures are restated from their published definitions
(``bleu1_precision``, ``rouge_f``: PARITY UNPINNED, anchored by hand-computed cases in tests/test_hostlogic_cpu.py); METEOR's exact
and Porter-stem matching stages are built (``meteor``, round 4), its WordNet-synonym stage is not (no WordNet data offline)."""
import collections
import re

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


# ------------------------------------------------------------------------------------------------- shared generate driver
def _generate_users(model, samples, tokenizer, image_preprocessor, device, users_per_batch, on_user, **gen_kw):
    """decode ``users_per_batch`` samples per generate() call; on_user(sample, texts) gets the decoded hypotheses of one user."""
    model.eval()
    group = []
    K = gen_kw.get("num_return_sequences", 1)

    def flush():
        if not group:
            return
        imgs = []
        for s in group:
            im = s["net_input"]["patch_images"]
            if isinstance(im, (list, tuple)):
                if image_preprocessor is None:
                    raise ValueError("raw images need an image_preprocessor")
                im = image_preprocessor(im)
            imgs.append(im.to(device=device, dtype=torch.bfloat16))
        if len({tuple(i.shape) for i in imgs}) != 1:
            raise ValueError("users decoded together must carry the same number of images (users_per_batch=1 otherwise)")
        vision_x = torch.stack(imgs).unsqueeze(2)                                        # (users, T, 1, 3, H, W)
        lens = [s["net_input"]["input_ids"].numel() for s in group]
        L = max(lens)
        pad = tokenizer.pad_token_id if tokenizer.pad_token_id is not None else tokenizer.eos_token_id
        ids = torch.full((len(group), L), pad, dtype=torch.long)
        for i, s in enumerate(group):
            ids[i, :lens[i]] = s["net_input"]["input_ids"].reshape(-1)
        mask = (torch.arange(L)[None, :] < torch.tensor(lens)[:, None]).long()
        gen = model.generate(vision_x=vision_x, lang_x=ids.to(device), attention_mask=mask.to(device),
                             eos_token_id=tokenizer.eos_token_id, pad_token_id=tokenizer.eos_token_id, **gen_kw)
        texts = tokenizer.batch_decode(gen, skip_special_tokens=True)
        for i, s in enumerate(group):
            on_user(s, texts[i * K:(i + 1) * K])
        group.clear()

    for s in samples:
        group.append(s)
        if len(group) == users_per_batch:
            flush()
    flush()


def _mean_rows(rows):
    return {k: float(np.mean([r[k] for r in rows])) for k in rows[0]} if rows else {}


@torch.no_grad()
def eval_model_search(model, samples, tokenizer, K=10, max_new_tokens=20, image_preprocessor=None, device="cuda", users_per_batch=1):
    """eval_search.py:29-175 -- eval_rec's ranking metrics on the search task (20 new tokens, no_repeat_ngram_size 0)."""
    rows = []
    _generate_users(model, samples, tokenizer, image_preprocessor, device, users_per_batch,
                    lambda s, texts: rows.append(user_metrics(relevance(texts, s["net_output"]["output_ids"], K), K)),
                    num_beams=K, num_return_sequences=K, early_stopping=True, max_new_tokens=max_new_tokens, no_repeat_ngram_size=0)
    return _mean_rows(rows)


# ---- explanation -------------------------------------------------------------------------------------------------------
def parse_rating_explanation(text):
    """eval_exp.py:116-125: words after the last "?"; the first one's suffix after "_" is the rating (3.0 if it does not parse),
    the rest the explanation ("Empty" when none)."""
    words = text.split("?")[-1].strip().split()
    try:
        rate = float(words[0].split("_")[-1])
    except (IndexError, ValueError):
        rate = 3.0
    exp = " ".join(words[1:])
    return rate, ("Empty" if exp == "" else exp)


def _tok13a(s):
    """sacreBLEU-style "13a" tokenisation as evaluate's bleu uses it (tokenizer_13a): punctuation split off, whitespace split."""
    s = s.replace("<skipped>", "").replace("-\n", "").replace("\n", " ")
    s = s.replace("&quot;", '"').replace("&amp;", "&").replace("&lt;", "<").replace("&gt;", ">")
    s = f" {s} "
    s = re.sub(r"([\{-\~\[-\` -\&\(-\+\:-\@\/])", r" \1 ", s)
    s = re.sub(r"([^0-9])([\.,])", r"\1 \2 ", s)
    s = re.sub(r"([\.,])([^0-9])", r" \1 \2", s)
    s = re.sub(r"([0-9])(-)", r"\1 \2 ", s)
    return s.split()


def bleu1_precision(predictions, references):
    """corpus-level modified unigram precision = evaluate bleu's ``precisions[0]`` (eval_exp.py:139): clipped unigram matches
    summed over the corpus / predicted unigrams summed over the corpus; one reference per prediction."""
    match = total = 0
    for p, r in zip(predictions, references):
        pc, rc = collections.Counter(_tok13a(p)), collections.Counter(_tok13a(r))
        match += sum(min(c, rc[w]) for w, c in pc.items())
        total += sum(pc.values())
    return match / total if total else 0.0


def _rouge_tokens(s):
    """rouge_score's default tokenizer without stemming (evaluate rouge: use_stemmer=False): lower-case, non-alphanumerics -> space."""
    return re.sub(r"[^a-z0-9]+", " ", s.lower()).split()


def _lcs(a, b):
    prev = [0] * (len(b) + 1)
    for x in a:
        cur = [0]
        for j, y in enumerate(b):
            cur.append(prev[j] + 1 if x == y else max(prev[j + 1], cur[j]))
        prev = cur
    return prev[-1]


def rouge_f(predictions, references):
    """mean over the pairs of the ROUGE-1 / ROUGE-2 / ROUGE-L F-measures (rouge_score definitions: n-gram overlap with clipped
    counts, sentence-level LCS for L; F = 2PR / (P + R)); evaluate's rouge aggregates with a bootstrap whose mid value is this
    mean up to resampling noise."""
    out = {"rouge1": [], "rouge2": [], "rougeL": []}

    def f(m, np_, nr):
        if m == 0 or np_ == 0 or nr == 0:
            return 0.0
        p, r = m / np_, m / nr
        return 2 * p * r / (p + r)
    for p, r in zip(predictions, references):
        tp, tr = _rouge_tokens(p), _rouge_tokens(r)
        for n, key in ((1, "rouge1"), (2, "rouge2")):
            gp = collections.Counter(tuple(tp[i:i + n]) for i in range(len(tp) - n + 1))
            gr = collections.Counter(tuple(tr[i:i + n]) for i in range(len(tr) - n + 1))
            out[key].append(f(sum(min(c, gr[g]) for g, c in gp.items()), sum(gp.values()), sum(gr.values())))
        out["rougeL"].append(f(_lcs(tp, tr), len(tp), len(tr)))
    return {k: float(np.mean(v)) if v else 0.0 for k, v in out.items()}


# ---- METEOR (eval_exp.py:65,146: evaluate's "meteor" = nltk.translate.meteor_score with alpha 0.9, beta 3, gamma 0.5) ---------------
# nltk aligns hypothesis and reference words in three stages: exact match, Porter-stem match, WordNet-synonym match.  The first two are
# restated here (Porter's 1980 algorithm; nltk / WordNet are not in the image and not under /root/reference, so: PARITY UNPINNED, and a
# LOWER bound of nltk's score whenever a synonym pair would have matched).  Per pair: m matched unigrams, P = m / |hyp|, R = m / |ref|,
# F = P R / (alpha P + (1 - alpha) R), penalty = gamma (chunks / m)^beta, score = F (1 - penalty); corpus value = mean over the pairs.
def _porter_stem(w):
    """M. F. Porter, "An algorithm for suffix stripping" (1980), steps 1a-5b; w lower-case."""
    if len(w) <= 2:
        return w
    vow = "aeiou"

    def cons(s, i):
        return s[i] not in vow and not (s[i] == "y" and i > 0 and cons(s, i - 1))

    def m(s):
        n, i, L = 0, 0, len(s)
        while i < L and cons(s, i):
            i += 1
        while i < L:
            while i < L and not cons(s, i):
                i += 1
            if i >= L:
                break
            n += 1
            while i < L and cons(s, i):
                i += 1
        return n

    def has_vowel(s):
        return any(not cons(s, i) for i in range(len(s)))

    def dbl(s):
        return len(s) >= 2 and s[-1] == s[-2] and cons(s, len(s) - 1)

    def cvc(s):
        return len(s) >= 3 and cons(s, len(s) - 3) and not cons(s, len(s) - 2) and cons(s, len(s) - 1) and s[-1] not in "wxy"
    if w.endswith("sses"):
        w = w[:-2]
    elif w.endswith("ies"):
        w = w[:-2]
    elif not w.endswith("ss") and w.endswith("s"):
        w = w[:-1]
    again = False
    if w.endswith("eed"):
        if m(w[:-3]) > 0:
            w = w[:-1]
    elif w.endswith("ed") and has_vowel(w[:-2]):
        w, again = w[:-2], True
    elif w.endswith("ing") and has_vowel(w[:-3]):
        w, again = w[:-3], True
    if again:
        if w.endswith(("at", "bl", "iz")):
            w += "e"
        elif dbl(w) and w[-1] not in "lsz":
            w = w[:-1]
        elif m(w) == 1 and cvc(w):
            w += "e"
    if w.endswith("y") and has_vowel(w[:-1]):
        w = w[:-1] + "i"
    for suf, rep in (("ational", "ate"), ("tional", "tion"), ("enci", "ence"), ("anci", "ance"), ("izer", "ize"), ("abli", "able"),
                     ("alli", "al"), ("entli", "ent"), ("eli", "e"), ("ousli", "ous"), ("ization", "ize"), ("ation", "ate"), ("ator", "ate"),
                     ("alism", "al"), ("iveness", "ive"), ("fulness", "ful"), ("ousness", "ous"), ("aliti", "al"), ("iviti", "ive"),
                     ("biliti", "ble")):
        if w.endswith(suf):
            if m(w[:-len(suf)]) > 0:
                w = w[:-len(suf)] + rep
            break
    for suf, rep in (("icate", "ic"), ("ative", ""), ("alize", "al"), ("iciti", "ic"), ("ical", "ic"), ("ful", ""), ("ness", "")):
        if w.endswith(suf):
            if m(w[:-len(suf)]) > 0:
                w = w[:-len(suf)] + rep
            break
    for suf in ("al", "ance", "ence", "er", "ic", "able", "ible", "ant", "ement", "ment", "ent", "ion", "ou", "ism", "ate", "iti", "ous",
                "ive", "ize"):
        if w.endswith(suf):
            stem = w[:-len(suf)]
            if m(stem) > 1 and (suf != "ion" or (stem and stem[-1] in "st")):
                w = stem
            break
    if w.endswith("e"):
        stem = w[:-1]
        if m(stem) > 1 or (m(stem) == 1 and not cvc(stem)):
            w = stem
    if m(w) > 1 and dbl(w) and w.endswith("l"):
        w = w[:-1]
    return w


def _meteor_align(hyp, ref):
    """nltk's greedy staged alignment: exact matches first, then stems, each stage walking the hypothesis from its END and taking the LAST
    still-unmatched equal reference word (meteor_score._match_enums); returns the sorted list of (hyp index, ref index) pairs."""
    pairs, uh, ur = [], list(enumerate(hyp)), list(enumerate(ref))
    for key in (lambda x: x, _porter_stem):
        i = len(uh) - 1
        while i >= 0:
            j = len(ur) - 1
            while j >= 0:
                if key(uh[i][1]) == key(ur[j][1]):
                    pairs.append((uh[i][0], ur[j][0]))
                    uh.pop(i)
                    ur.pop(j)
                    break
                j -= 1
            i -= 1
    return sorted(pairs)


def meteor(predictions, references, alpha=0.9, beta=3.0, gamma=0.5):
    """mean METEOR of the pairs (exact + stem stages; module comment above)."""
    scores = []
    for p, r in zip(predictions, references):
        hyp, ref = _tok13a(p.lower()), _tok13a(r.lower())
        al = _meteor_align(hyp, ref)
        n = len(al)
        if n == 0 or not hyp or not ref:
            scores.append(0.0)
            continue
        prec, rec = n / len(hyp), n / len(ref)
        fmean = prec * rec / (alpha * prec + (1 - alpha) * rec)
        chunks = 1 + sum(1 for (h0, r0), (h1, r1) in zip(al[:-1], al[1:]) if not (h1 == h0 + 1 and r1 == r0 + 1))
        scores.append(fmean * (1 - gamma * (chunks / n) ** beta))
    return float(np.mean(scores)) if scores else 0.0


@torch.no_grad()
def eval_model_exp(model, samples, tokenizer, max_new_tokens=256, num_beams=5, image_preprocessor=None, device="cuda", users_per_batch=1):
    """eval_exp.py:31-205: rating + explanation generation.  Returns mae, rmse, bleu (unigram precision), rouge1 / rouge2 / rougeL and
    ``meteor_exact_stem``.  The last one is NOT the reference's "meteor" and must not be compared with eval_exp.py numbers: evaluate's
    metric tokenises with nltk ``word_tokenize``, stems with nltk's PorterStemmer in its NLTK_EXTENSIONS mode and adds a WordNet synonym
    stage; here the exact + (1980 Porter) stem stages run on whitespace tokens (no nltk / WordNet data offline) -- a lower bound with
    another tokenisation, hence another key.  ``unpinned_metrics`` lists the keys whose definition is a restatement
    without a pin on the reference's library."""
    abs_err, sq_err, gen_exps, real_exps = [], [], [], []

    def on_user(s, texts):
        rate, exp = parse_rating_explanation(texts[0])
        real = float(np.asarray(s["net_output"]["output_ratings"], dtype=np.float64).reshape(-1)[0])
        abs_err.append(abs(rate - real))
        sq_err.append((rate - real) ** 2)
        gen_exps.append(exp)
        real_exps.append(s["net_output"]["output_exps"][0])
    _generate_users(model, samples, tokenizer, image_preprocessor, device, users_per_batch, on_user,
                    num_beams=num_beams, num_return_sequences=1, early_stopping=True, max_new_tokens=max_new_tokens)
    if not abs_err:
        return {}
    out = {"mae": float(np.mean(abs_err)), "rmse": float(np.sqrt(np.mean(sq_err))), "bleu": bleu1_precision(gen_exps, real_exps)}
    out.update(rouge_f(gen_exps, real_exps))
    out["meteor_exact_stem"] = meteor(gen_exps, real_exps)
    out["unpinned_metrics"] = ["bleu", "rouge1", "rouge2", "rougeL", "meteor_exact_stem"]
    return out


# ---- item selection ----------------------------------------------------------------------------------------------------
def selection_scores(text, output_ids):
    """eval_img_sel.py:96-113: the SET of generated words after the last "?" against the right selections "s_i"."""
    gen = set(text.split("?")[-1].strip().split())
    gts = [f"s_{i}" for i in np.asarray(output_ids).reshape(-1)]
    r = float(np.sum([1 if g in gts else 0 for g in gen], dtype=np.float64))
    recall = r / len(gts)
    precision = 0 if len(gen) == 0 else r / len(gen)
    f1 = (2.0 * precision * recall) / (precision + recall) if (precision > 0 or recall > 0) else 0.0
    return {"recall": recall, "precision": precision, "f1": f1}


@torch.no_grad()
def eval_model_img_sel(model, samples, tokenizer, max_new_tokens=40, num_beams=2, image_preprocessor=None, device="cuda", users_per_batch=1):
    """eval_img_sel.py:29-148."""
    rows = []
    _generate_users(model, samples, tokenizer, image_preprocessor, device, users_per_batch,
                    lambda s, texts: rows.append(selection_scores(texts[0], s["net_output"]["output_ids"])),
                    num_beams=num_beams, num_return_sequences=1, early_stopping=True, max_new_tokens=max_new_tokens, no_repeat_ngram_size=0)
    return _mean_rows(rows)


# ---- image-token generation --------------------------------------------------------------------------------------------
@torch.no_grad()
def eval_model_img_gen(model, samples, tokenizer, max_new_tokens=600, image_preprocessor=None, device="cuda", users_per_batch=1):
    """eval_img_gen.py:29-156: greedy decode of the next item's image-token sequence (600 new tokens); returns
    {item id: [generated text after the last "?"]} -- what the reference dumps to save_img_gen/*.json -- plus, beside it,
    the share of users whose text equals the target string (the reference's commented-out hit metric)."""
    texts_dict, hits = {}, []

    def on_user(s, texts):
        texts = [t.split("</s>")[0] for t in texts]
        texts = [t.split("?")[-1].strip() for t in texts]
        item = s["net_output"].get("items", len(texts_dict))
        item = item.item() if hasattr(item, "item") else (item[0] if isinstance(item, (list, tuple)) else item)
        texts_dict[item] = texts
        hits.append(float(texts[0] == s["net_output"]["output_ids"]))
    _generate_users(model, samples, tokenizer, image_preprocessor, device, users_per_batch, on_user,
                    num_beams=1, num_return_sequences=1, early_stopping=True, max_new_tokens=max_new_tokens)
    return {"texts": texts_dict, "exact": float(np.mean(hits)) if hits else 0.0}
