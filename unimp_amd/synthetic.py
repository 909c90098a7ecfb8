"""Synthetic batches with the reference's layout: the tensors collate_fn hands to
train_one_epoch (UniMP/pipeline/mm_utils/collate_rec.py:38-74, UniMP/mmrec.py:135-141) built from the prompt
template of rec_dataset.py:414-424:

    [BOS] ( <image> text... <answer> item_k <|endofchunk|> ) x T   question... <answer> item_next [EOS]   [PAD]...

Pure data plumbing (CPU, seeded); identical generator for the HIP path, the oracle and the CPU baseline.
"""
import torch


class TokenLayout:
    """ids of the special tokens for a vocabulary laid out as UniMP builds it (mmrec.py:538-581):
    base vocab | <|endofchunk|> <image> <PAD> | <answer> rate_1..5 s_0..4 | item_0..N-1 | img_0..1023"""

    def __init__(self, base_vocab=50277, n_items=22738, n_img_tokens=1024):
        self.base_vocab = base_vocab
        self.bos = self.eos = 0
        self.eoc, self.media, self.pad = base_vocab, base_vocab + 1, base_vocab + 2
        self.answer = base_vocab + 3
        self.item0 = base_vocab + 3 + 1 + 5 + 5
        self.n_items = n_items
        self.vocab = self.item0 + n_items + n_img_tokens

    def special(self):
        return dict(answer_id=self.answer, eoc_id=self.eoc, pad_id=self.pad, media_id=self.media)


def make_batch(layout, batch, T, L, image_size=224, seed=1234, weight=2.0, min_fill=0.75, device=None, vision_dtype=torch.float32):
    """returns dict(vision_x (b,T,1,3,H,W), lang_x (b,L) int64, attention_mask (b,L) int64, weights (b,) fp32)."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.full((batch, L), layout.pad, dtype=torch.int64)
    mask = torch.zeros((batch, L), dtype=torch.int64)
    q_len = 12
    fixed = 1 + T * 4 + q_len + 3                       # BOS + per chunk (<image>,<answer>,item,<eoc>) + question + <answer> item EOS
    for b in range(batch):
        real = int(torch.randint(int(min_fill * L), L + 1, (1,), generator=g))
        n_text = max(T, real - fixed)
        per = [n_text // T + (1 if i < n_text % T else 0) for i in range(T)]
        s = [layout.bos]
        for t in range(T):
            s.append(layout.media)
            s += torch.randint(1, layout.base_vocab, (per[t],), generator=g).tolist()
            s += [layout.answer, layout.item0 + int(torch.randint(0, layout.n_items, (1,), generator=g)), layout.eoc]
        s += torch.randint(1, layout.base_vocab, (q_len,), generator=g).tolist()
        s += [layout.answer, layout.item0 + int(torch.randint(0, layout.n_items, (1,), generator=g)), layout.eos]
        s = s[:L]
        ids[b, :len(s)] = torch.tensor(s)
        mask[b, :len(s)] = 1
    vis = torch.randn((batch, T, 1, 3, image_size, image_size), generator=g).to(vision_dtype)
    w = torch.full((batch,), float(weight))
    out = dict(vision_x=vis, lang_x=ids, attention_mask=mask, weights=w)
    if device is not None:
        out = {k: v.to(device) for k, v in out.items()}
    return out


def make_exp_batch(layout, batch, T, L, image_size=224, seed=1234, weight=1.0, exp_len=28, device=None, vision_dtype=torch.float32):
    """rating + explanation samples (BASELINE config 3's "explain" task; rec_dataset.py:1100-1134): T - 1 history chunks
    ``<image> meta... <answer> rate_r explanation... <|endofchunk|>`` and the query chunk ``<image> meta... question... <answer> rate_r
    explanation... [EOS]`` -- the label mask (mmrec.py:143-168) keeps every token between an ``<answer>`` and the chunk's end, so a sample
    carries T x (1 + exp_len) + 1 labeled positions (233 at T = 8, exp_len = 28) where the rec template carries T + 2; loss weight 1.0
    (rec_dataset.py:452: 2.0 is the rec task's).  Same return layout as make_batch."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.full((batch, L), layout.pad, dtype=torch.int64)
    mask = torch.zeros((batch, L), dtype=torch.int64)
    rate0 = layout.answer + 1                           # rate_1 .. rate_5 follow <answer> in the vocabulary (TokenLayout)
    q_len = 10
    fixed = 1 + T * (3 + 1 + exp_len) + q_len           # BOS + per chunk (<image>, <answer>, rate, explanation, <eoc> / EOS) + the question
    for b in range(batch):
        n_meta = max(T, L - fixed - int(torch.randint(0, max(1, L // 16), (1,), generator=g)))
        per = [n_meta // T + (1 if i < n_meta % T else 0) for i in range(T)]
        text = lambda n: torch.randint(1, layout.base_vocab, (n,), generator=g).tolist()
        s = [layout.bos]
        for t in range(T):
            s += [layout.media] + text(per[t])
            if t == T - 1:
                s += text(q_len)
            s += [layout.answer, rate0 + int(torch.randint(0, 5, (1,), generator=g))] + text(exp_len)
            s += [layout.eoc if t < T - 1 else layout.eos]
        if len(s) > L:
            raise ValueError(f"explain sample of {len(s)} tokens does not fit L = {L}")
        ids[b, :len(s)] = torch.tensor(s)
        mask[b, :len(s)] = 1
    vis = torch.randn((batch, T, 1, 3, image_size, image_size), generator=g).to(vision_dtype)
    out = dict(vision_x=vis, lang_x=ids, attention_mask=mask, weights=torch.full((batch,), float(weight)))
    if device is not None:
        out = {k: v.to(device) for k, v in out.items()}
    return out


def make_imggen_batch(layout, batch, T=2, L=1024, image_size=224, seed=5, device=None, vision_dtype=torch.float32):
    """image-token generation samples (BASELINE config 5's task; rec_dataset.py:613-664, eval_img_gen.py:102-111): T history chunks
    ``<image> title... ID img_a,img_b,...(256 VQGAN code tokens) <|endofchunk|>``, then the query and ``<answer>`` + the target item's
    256 code tokens + EOS; only that final span is labeled (257 positions); loss weight 1.0; ~860 real tokens padded to L."""
    g = torch.Generator().manual_seed(seed)
    img0 = layout.item0 + layout.n_items
    ids = torch.full((batch, L), layout.pad, dtype=torch.int64)
    mask = torch.zeros((batch, L), dtype=torch.int64)
    for b in range(batch):
        codes = lambda: (img0 + torch.randint(0, 1024, (256,), generator=g)).tolist()
        text = lambda n: torch.randint(1, layout.base_vocab, (n,), generator=g).tolist()
        s = [layout.bos]
        for _ in range(T):
            s += [layout.media] + text(20) + codes() + [layout.eoc]
        s += text(40) + [layout.answer] + codes() + [layout.eos]
        if len(s) > L:
            raise ValueError(f"img-gen sample of {len(s)} tokens does not fit L = {L}")
        ids[b, :len(s)] = torch.tensor(s)
        mask[b, :len(s)] = 1
    vis = torch.randn((batch, T, 1, 3, image_size, image_size), generator=g).to(vision_dtype)
    out = dict(vision_x=vis, lang_x=ids, attention_mask=mask, weights=torch.ones(batch))
    if device is not None:
        out = {k: v.to(device) for k, v in out.items()}
    return out
