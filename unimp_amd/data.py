"""Input pipeline either side of the training step.

* ``ImagePreprocessor``: the reference's ``patch_resize_transform`` (UniMP/pipeline/mm_utils/rec_dataset.py:91-107:
  bicubic resize to 224 x 224 via Pillow, ToTensor, Normalize with the CLIP statistics of rec_dataset.py:30-31) on the GPU:
  the host hands over DECODED uint8 RGB images of any size (JPEG decoding stays on the host), the resize / scale /
  normalise / cast run as HIP kernels (csrc/preprocess.hip), bit-exact with Pillow's 8-bit resampler.  Instead of
  4.8 MB of fp32 pixels per sample the PCIe link carries the raw bytes.
* ``collate_fn`` / ``collate_tokens``: UniMP/pipeline/mm_utils/collate_rec.py:38-115 (right padding with pad_idx, attention
  masks padded with 0, per-sample loss weights, stacked images).
* ``rec_prompt``: the text side of ``process_train_rec_pair`` (rec_dataset.py:372-437).
"""
import ctypes as C
import numpy as np
import torch

from . import _lib
from . import ops

FLAMINGO_MEAN = (0.48145466, 0.4578275, 0.40821073)      # rec_dataset.py:30
FLAMINGO_STD = (0.26862954, 0.26130258, 0.27577711)      # rec_dataset.py:31
_PBITS = 22                                               # Pillow: PRECISION_BITS = 32 - 8 - 2


def bicubic_taps(in_size, out_size):
    """Pillow's tap table for one axis (libImaging/Resample.c precompute_coeffs + normalize_coeffs_8bpc, bicubic a = -0.5),
    vectorised over the output coordinate with the SAME double-precision operation order, so the 22-bit integers are
    Pillow's.  -> int32 [out_size, 2 + ksize]: (first source index, tap count, taps...)."""
    scale = in_size / out_size
    fs = max(scale, 1.0)
    support = 2.0 * fs
    ksize = int(np.ceil(support)) * 2 + 1
    ss = 1.0 / fs
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.trunc(center - support + 0.5).astype(np.int64)
    xmin = np.maximum(xmin, 0)
    xmax = np.trunc(center + support + 0.5).astype(np.int64)
    xmax = np.minimum(xmax, in_size) - xmin
    k = np.zeros((out_size, ksize), dtype=np.float64)
    ww = np.zeros(out_size, dtype=np.float64)
    a = -0.5
    for x in range(ksize):
        t = np.abs((x + xmin - center + 0.5) * ss)
        w = np.where(t < 1.0, ((a + 2.0) * t - (a + 3.0)) * t * t + 1, np.where(t < 2.0, (((t - 5) * t + 8) * t - 4) * a, 0.0))
        w = np.where(x < xmax, w, 0.0)
        k[:, x] = w
        ww = ww + w                                   # same left-to-right accumulation as the C loop
    nz = ww != 0.0
    k[nz] = k[nz] / ww[nz, None]
    fixed = np.trunc(k * float(1 << _PBITS) + np.where(k < 0, -0.5, 0.5)).astype(np.int32)
    fixed[np.arange(ksize)[None, :] >= xmax[:, None]] = 0
    out = np.empty((out_size, 2 + ksize), dtype=np.int32)
    out[:, 0], out[:, 1], out[:, 2:] = xmin, xmax, fixed
    return out


class _ImageDesc(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("src_off", "H", "W", "kx_off", "ky_off", "ksx", "ksy", "tmp_off")]


class _Slot:
    """one stage of the H2D ring: pinned host bytes, their device image, and the two events that order its reuse."""

    def __init__(self):
        self.host = self.dev = None
        self.h2d_done = torch.cuda.Event()
        self.kernel_done = torch.cuda.Event()

    def reserve(self, nbytes, device):
        """make room for nbytes; True if the device buffer was (re)allocated.  A fresh block comes from the caching allocator's pool of
        the CURRENT (compute) stream and may be one the host has just freed while kernels that use it are still queued there: the
        caller must order the copy stream behind the compute stream before the first copy into it."""
        if self.host is None or self.host.numel() < nbytes:
            cap = max(nbytes, 2 * (self.host.numel() if self.host is not None else 0), 1 << 20)
            self.host = torch.empty(cap, dtype=torch.uint8, pin_memory=True)
            self.dev = torch.empty(cap, dtype=torch.uint8, device=device)
            return True
        return False


class PendingImages:
    """a batch whose bytes are on their way to the device (ImagePreprocessor.submit); ``get()`` enqueues the kernels."""

    def __init__(self, owner, slot, n, max_h, offs, want_u8):
        self.owner, self.slot, self.n, self.max_h, self.offs, self.want_u8 = owner, slot, n, max_h, offs, want_u8
        self._out = None

    def get(self):
        if self._out is not None:
            return self._out
        o, S = self.owner, self.owner.size
        out = torch.empty((self.n, 3, S, S), dtype=o.dtype, device=o.device)
        u8 = torch.empty((self.n, S, S, 3), dtype=torch.uint8, device=o.device) if self.want_u8 else None
        if self.n:
            sl = self.slot
            cur = torch.cuda.current_stream(o.device)
            cur.wait_event(sl.h2d_done)                     # the compute stream waits for this batch's copy only
            src_off, tab_off, dd_off, tmp_bytes = self.offs
            base = sl.dev.data_ptr()
            tmp = torch.empty(max(tmp_bytes, 1), dtype=torch.uint8, device=o.device)
            _lib.check(_lib.lib().unimp_image_resize_normalize(
                base + src_off, base + dd_off, self.n, self.max_h, base + tab_off, tmp.data_ptr(), S, S,
                C.cast(o._mean, C.c_void_p), C.cast(o._std, C.c_void_p), out.data_ptr(), int(o.dtype == torch.float32),
                u8.data_ptr() if u8 is not None else None, ops._stream()), "image_resize_normalize")
            sl.kernel_done.record(cur)                      # the slot's device bytes may be overwritten after this
        self._out = (out, u8) if self.want_u8 else out
        return self._out


class ImagePreprocessor:
    """callable: list of decoded RGB images (uint8 [H, W, 3] numpy arrays / torch tensors / PIL images, any sizes) ->
    [n, 3, size, size] tensor on `device` (bf16 by default, like ``images.to(device, dtype=cast_dtype)`` at mmrec.py:135).

    Host -> device: the decoded bytes, the tap tables and the per-image descriptors are packed straight into ONE pinned host
    buffer and cross PCIe as one asynchronous copy on a dedicated copy stream; a ring of ``depth`` (2) such stages lets
    ``submit(next_batch)`` pack and copy batch i + 1 while the GPU still computes on batch i (the reference hides the same work
    behind 4 DataLoader workers per rank, mmrec.py:403).  ``__call__`` = ``submit(...).get()``."""

    def __init__(self, size=224, mean=FLAMINGO_MEAN, std=FLAMINGO_STD, device="cuda", dtype=torch.bfloat16, depth=2):
        if dtype not in (torch.bfloat16, torch.float32):
            raise ValueError("ImagePreprocessor: dtype must be bfloat16 or float32")
        self.size, self.device, self.dtype = size, torch.device(device), dtype
        self._mean = (C.c_float * 3)(*mean)
        self._std = (C.c_float * 3)(*std)
        self._taps = {}                      # in_size -> int32 table (host), shared by both axes
        self._ring, self._next, self._depth, self._copy_stream = None, 0, depth, None

    def _table(self, n):
        t = self._taps.get(n)
        if t is None:
            t = self._taps[n] = bicubic_taps(n, self.size)
        return t

    @staticmethod
    def _as_u8(img):
        if isinstance(img, torch.Tensor):
            img = img.cpu().numpy()
        a = np.asarray(img if isinstance(img, np.ndarray) else img.convert("RGB"))
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
            raise ValueError(f"ImagePreprocessor: expected a decoded RGB uint8 [H, W, 3] image, got {a.dtype} {a.shape}")
        return a

    def submit(self, images, return_u8=False):
        """pack + start the asynchronous H2D copy of a batch; returns a PendingImages (``.get()`` -> tensor)."""
        S = self.size
        imgs = [self._as_u8(i) for i in images]
        n = len(imgs)
        if n == 0:
            return PendingImages(self, None, 0, 0, None, return_u8)
        if self.device.type != "cuda":
            raise _lib.UnimpHipError("ImagePreprocessor needs a HIP device (no CPU fallback exists)")
        if self._ring is None:
            self._ring = [_Slot() for _ in range(self._depth)]
            self._copy_stream = torch.cuda.Stream(self.device)
        descs = (_ImageDesc * n)()
        tabs, tab_off, src_off, tmp_off, tpos = [], {}, 0, 0, 0
        for i, a in enumerate(imgs):
            H, W = a.shape[:2]
            d = descs[i]
            d.src_off, d.H, d.W, d.tmp_off = src_off, H, W, tmp_off
            for axis, nn in (("x", W), ("y", H)):
                if nn == S:
                    ks, off = 0, 0
                else:
                    if nn not in tab_off:
                        t = self._table(nn)
                        tab_off[nn] = (tpos, t.shape[1] - 2)
                        tabs.append(t.reshape(-1))
                        tpos += t.size
                    off, ks = tab_off[nn]
                if axis == "x":
                    d.kx_off, d.ksx = off, ks
                else:
                    d.ky_off, d.ksy = off, ks
            src_off += H * W * 3
            tmp_off += H * S * 3
        # one pinned image: [ pixels | pad to 16 | tap tables (int32) | pad to 16 | descriptors (int64) ]
        r16 = lambda x: (x + 15) // 16 * 16
        t_off = r16(src_off)
        t_bytes = max(tpos, 1) * 4
        d_off = r16(t_off + t_bytes)
        total = d_off + C.sizeof(descs)
        sl = self._ring[self._next]
        self._next = (self._next + 1) % self._depth
        sl.h2d_done.synchronize()                            # the previous copy OUT of this pinned buffer has finished (normally long ago)
        fresh = sl.reserve(total, self.device)
        hb = sl.host.numpy()
        pos = 0
        for a in imgs:                                       # packed straight into pinned memory: no concatenate, no pageable staging
            k = a.size
            hb[pos:pos + k] = a.reshape(-1)
            pos += k
        if tabs:
            hb[t_off:t_off + tpos * 4] = np.concatenate(tabs).view(np.uint8)
        hb[d_off:d_off + C.sizeof(descs)] = np.frombuffer(descs, dtype=np.uint8)
        cs = self._copy_stream
        cs.wait_event(sl.kernel_done)                        # the kernels that read this slot's previous device bytes are done
        if fresh:
            # first use / regrow: kernel_done says nothing about this block.  Whatever the compute stream still has queued on the block's
            # previous owner must finish before the copy writes it (and must not write over the copied bytes afterwards); the block is
            # also used on the copy stream from now on, which the allocator has to know before it hands the block out again.
            cs.wait_stream(torch.cuda.current_stream(self.device))
            sl.dev.record_stream(cs)
        with torch.cuda.stream(cs):
            sl.dev[:total].copy_(sl.host[:total], non_blocking=True)
            sl.h2d_done.record(cs)
        return PendingImages(self, sl, n, max(a.shape[0] for a in imgs), (0, t_off, d_off, tmp_off), return_u8)

    def __call__(self, images, return_u8=False):
        return self.submit(images, return_u8).get()


# ------------------------------------------------------------------------------------------------ collate (host logic)
def collate_tokens(values, pad_idx, eos_idx=None, left_pad=False, move_eos_to_beginning=False, pad_to_length=None,
                   pad_to_multiple=1, pad_to_bsz=None):
    """list of 1-D (or 2-D) tensors -> one padded tensor (collate_rec.py:77-115)."""
    size = max(v.size(0) for v in values)
    size = size if pad_to_length is None else max(size, pad_to_length)
    if pad_to_multiple != 1 and size % pad_to_multiple != 0:
        size = int(((size - 0.1) // pad_to_multiple + 1) * pad_to_multiple)
    if values[0].dim() == 1:
        res = values[0].new_full((len(values), size), pad_idx)
    elif values[0].dim() == 2:
        if move_eos_to_beginning:
            raise AssertionError("move_eos_to_beginning needs 1-D inputs")
        res = values[0].new_full((len(values), size, values[0].size(1)), pad_idx)
    else:
        raise NotImplementedError
    for i, v in enumerate(values):
        dst = res[i][size - len(v):] if left_pad else res[i][:len(v)]
        if move_eos_to_beginning:
            dst[0] = v[-1] if eos_idx is None else eos_idx
            dst[1:] = v[:-1]
        else:
            dst.copy_(v)
    return res


def collate_fn(samples, pad_idx, eos_idx):
    """collate_rec.py:38-74: samples are ``{"net_input": {"input_ids", "attention_masks", "patch_images", "weights"}}``."""
    if len(samples) == 0:
        return {}
    ni = [s["net_input"] for s in samples]
    longest = max(x["input_ids"].size(0) for x in ni)
    batch = {"net_input": {
        "input_ids": collate_tokens([x["input_ids"] for x in ni], pad_idx, eos_idx=eos_idx, pad_to_length=longest),
        "attention_masks": collate_tokens([x["attention_masks"] for x in ni], 0, eos_idx=eos_idx, pad_to_length=longest),
        "weights": torch.tensor([x["weights"] for x in ni]),
    }}
    batch["net_input"]["patch_images"] = torch.stack([x["patch_images"] for x in ni], dim=0)
    return batch


def rec_prompt(history, target, question="What is the next item recommended to the user?"):
    """text of ``process_train_rec_pair`` (rec_dataset.py:395-424, naive item ids): history = [(item_id, meta_text), ...]."""
    s = "".join(f"<image> {meta} <answer> item_{item} <|endofchunk|> " for item, meta in history)
    return s + f"{question} <answer> item_{target}"


# ------------------------------------------------------------------------------------------------ dataset (host logic)
class RecDataset:
    """The sequential-recommendation task of UniMP's ``RecDataset`` (UniMP/pipeline/mm_utils/rec_dataset.py:57-297 for the
    on-disk layout, 300-370 for the item descriptions, 372-456 / 458-533 for the train / eval samples, 1260-1278 collate).

    Layout under ``folder``: ``{split}_users.json`` (``test_users.json`` for every non-train split) = {user: [[item, ...],
    ...]}; ``meta_{subset}.json``; ``{subset}/{item}.jpg``; ``id2semantic.json`` when ``use_semantic``.
    The numpy global RNG is consumed exactly like the reference does (one ``np.random.choice`` per training sample for the
    history window, one unused ``np.random.random()`` inside every item description), so a seeded run yields the same
    samples.  ``image_transform``: per-image host callable (the factory's ``image_processor``); with ``defer_images=True``
    the decoded uint8 arrays are kept instead and ``collate`` returns them under ``patch_images_raw`` for
    ``ImagePreprocessor`` to resize / normalise on the GPU."""

    HISTORY = {"all": 5, "netflix": 3, "hm": 8}
    QUESTION = "What is the next item recommended to the user?"

    ALL_ITEMS = {"all": 22738, "netflix": 1870, "hm": 14901}           # rec_dataset.py:274-279 (negatives of the selection task)

    def __init__(self, folder, subset, tokenizer, split="train", use_semantic=False, image_transform=None, defer_images=False,
                 task="rec", n_items=None):
        import json
        import os
        if subset not in self.HISTORY:
            raise ValueError(f"RecDataset: unknown subset {subset!r}")
        if task not in ("rec", "search", "exp", "img_sel", "img_gen"):
            raise NotImplementedError(f"RecDataset: unknown task {task!r}")
        self.task = task
        self.all_items = set(range(n_items if n_items is not None else self.ALL_ITEMS[subset]))
        if image_transform is None and not defer_images:
            raise ValueError("RecDataset: pass image_transform or defer_images=True")
        self.folder, self.subset, self.split, self.tokenizer = folder, subset, split, tokenizer
        self.use_semantic, self.transform, self.defer = use_semantic, image_transform, defer_images
        self.img_folder = os.path.join(folder, subset)
        self.history_len = 2 if (task == "img_gen" and subset == "all") else self.HISTORY[subset]          # rec_dataset.py:143-152
        with open(os.path.join(folder, f"meta_{subset}.json")) as f:
            self.meta_data = json.load(f)
        name = {"rec": None, "search": None, "exp": f"{split}_{subset}_exp.json", "img_sel": f"{split}_{subset}_img_sel.json",
                "img_gen": f"{split}_{subset}.json" if split == "train" else None}[task]
        with open(os.path.join(folder, name or (f"{split}_users.json" if split == "train" else "test_users.json"))) as f:
            self.data = json.load(f)
        self.seqs, self.keys = list(self.data.values()), list(self.data.keys())
        if task == "img_gen":        # image-token generation: item sequences from the retrieval file, targets = VQGAN code ids
            with open(os.path.join(folder, f"search_merge_{split}.txt")) as f:
                self.seqs = list(json.load(f))
            with open(os.path.join(folder, "img_id2semantic.json")) as f:
                self.img_id2semantic = json.load(f)
        if use_semantic:
            self.len_semanticid = 3
            with open(os.path.join(folder, "id2semantic.json")) as f:
                self.id2semantic = json.load(f)
        self.bos_item = torch.LongTensor([tokenizer.bos_token_id])
        self.eos_item = torch.LongTensor([tokenizer.eos_token_id])

    def __len__(self):
        return len(self.seqs)

    # rec_dataset.py:300-370
    def describe(self, item):
        cut = lambda s: " ".join(s.split()[:20])
        s = self.meta_data[str(item)]
        np.random.random()                                     # the reference draws (and ignores) p here
        if self.subset == "all":
            g = lambda k: "Unknown" if s[k] == "" else s[k]
            return f"Category {cut(g('category'))} Price {g('price')} Brand {cut(g('brand'))} Title {cut(g('title'))}"
        if self.subset == "netflix":
            return f"Title {cut(s[1])} Release Date {s[0]}"
        return f"Name {cut(s[0])} Appearance {cut(s[1])} Color {cut(s[2])} Section {cut(s[3])}"

    def _item_token(self, item):
        if not self.use_semantic:
            return f"item_{item}"
        ids = self.id2semantic[str(item)].split(",")
        sep = " " if self.task == "search" else ""             # rec_dataset.py:866 vs :411
        return sep.join(f"item_{v}" if i < self.len_semanticid else f"item_last_{v}" for i, v in enumerate(ids))

    def _image(self, item):
        import os
        from PIL import Image
        img = Image.open(os.path.join(self.img_folder, f"{item}.jpg")).convert("RGB")
        return np.asarray(img).copy() if self.defer else self.transform(img)

    def _tokenize(self, text):
        kw = {"truncation": True} if self.task == "rec" else {}          # the search task tokenises without truncation
        t = self.tokenizer(text, return_tensors="pt", add_special_tokens=False, **kw)
        return t["input_ids"].squeeze(0), t["attention_mask"].squeeze(0)

    def _search_item(self, index):
        """rec_dataset.py:842-913 (train) / 915-979 (eval): history as in rec, then "Query: <category of the target>"."""
        seq = [it[0] for it in self.seqs[index]]
        imgs, text = [], ""
        ask = "What is the related item ID to the query based on the history?"
        query = lambda item: self.meta_data[str(item)]["keywords" if self.subset == "cloth" else "category"]
        if self.split == "train":
            start = np.random.choice(list(range(0, len(seq) - self.history_len)), 1)[0]
            end = start + self.history_len
            for item in seq[start:end]:
                imgs.append(self._image(item))
                text += f"<image> {self.describe(item)} <answer> {self._item_token(item)} <|endofchunk|> "
            text += f"Query: {query(seq[end])} {ask} <answer> {self._item_token(seq[end])}"
            ids, mask = self._tokenize(text)
            one = torch.LongTensor([1])
            return {"net_input": {"input_ids": torch.cat([self.bos_item, ids, self.eos_item]), "attention_masks": torch.cat([one, mask, one]),
                                  "patch_images": imgs if self.defer else torch.stack(imgs, dim=0), "weights": torch.tensor(1.0)}}
        for item in seq[-5:-1]:
            imgs.append(self._image(item))
            text += f"<image> {self.describe(item)} {self._item_token(item)} <|endofchunk|> "
        text += f"Query: {query(seq[-1])} {ask} <answer>"
        ids, mask = self._tokenize(text)
        return {"net_input": {"input_ids": ids, "attention_masks": mask, "patch_images": imgs if self.defer else torch.stack(imgs, dim=0),
                              "input_len": len(text.split(" "))}, "net_output": {"output_ids": self._item_token(seq[-1])}}

    def _finish(self, text, imgs, weight=None, **net_output):
        ids, mask = self._tokenize(text)
        patch = imgs if self.defer else torch.stack(imgs, dim=0)
        if weight is not None:                                 # training sample: BOS + text + EOS
            one = torch.LongTensor([1])
            return {"net_input": {"input_ids": torch.cat([self.bos_item, ids, self.eos_item]), "attention_masks": torch.cat([one, mask, one]),
                                  "patch_images": patch, "weights": torch.tensor(weight)}}
        return {"net_input": {"input_ids": ids, "attention_masks": mask, "patch_images": patch, "input_len": len(text.split(" "))},
                "net_output": net_output}

    def _exp_item(self, index):
        """rating + explanation (rec_dataset.py:1100-1156 train, 1158-1212 eval); sequences of [item, text, rating]."""
        full, imgs, text = self.seqs[index], [], ""
        ask = "What is the rating and explanation for the item? <answer>"
        if self.split == "train":
            start = np.random.choice(list(range(0, len(full) - self.history_len + 1)), 1)[0]
            end = start + self.history_len - 1
            cut = lambda t: " ".join(t.split()[:30])
            for it in full[start:end]:
                imgs.append(self._image(it[0]))
                text += f"<image> {self.describe(it[0])} <answer> rate_{int(it[2])} {cut(it[1])} <|endofchunk|> "
            it = full[end]
            imgs.append(self._image(it[0]))
            return self._finish(text + f"<image> {self.describe(it[0])} {ask} rate_{int(it[2])} {cut(it[1])}", imgs, weight=1.0)
        for it in full[-5:-1]:
            imgs.append(self._image(it[0]))
            text += f"<image> {self.describe(it[0])} <answer> rate_{int(it[2])} {it[1]} <|endofchunk|> "
        it = full[-1]
        imgs.append(self._image(it[0]))
        return self._finish(text + f"<image> {self.describe(it[0])} {ask}", imgs, output_ratings=[int(it[2])], output_exps=[it[1]])

    def _img_sel_item(self, index):
        """item selection (rec_dataset.py:981-1046 train, 1048-1098 eval); the last element ends with (candidate set, indices
        of the right ones)."""
        full, imgs, text = self.seqs[index], [], "User history: "
        ask = "Can you select the suitable item from above for the user? <answer>"
        if self.split == "train":
            num_items, cur = 3, []
            for it in full[-(self.history_len - num_items + 1):-1]:
                cur.append(it[0])
                imgs.append(self._image(it[0]))
                text += f"<image> {self.describe(it[0])} <|endofchunk|> "
            text += "Select from: "
            cand = full[-1][-2]
            gt = [cand[i] for i in full[-1][-1]]
            cur += gt
            labels = np.random.choice(list(range(num_items)), len(gt), replace=False)
            neg_index = list(set(range(num_items)) - set(labels))
            negs = np.random.choice(list(self.all_items - set(cur)), num_items - len(gt), replace=False)
            chosen = [0] * num_items
            for i, item in enumerate(gt):
                chosen[labels[i]] = item
            for i, item in enumerate(negs):
                chosen[neg_index[i]] = item
            for i, item in enumerate(chosen):
                imgs.append(self._image(item))
                text += f"<image> Selection s_{i} {self.describe(item)} <|endofchunk|> "
            text += ask + " " + "".join(f"s_{l} " for l in labels)
            return self._finish(text, imgs, weight=1.0)
        for it in full[-5:-1]:
            imgs.append(self._image(it[0]))
            text += f"<image> {self.describe(it[0])} <|endofchunk|> "
        text += "Select from: "
        for i, item in enumerate(full[-1][-2]):
            imgs.append(self._image(item))
            text += f"<image> Selection s_{i} {self.describe(item)} <|endofchunk|> "
        return self._finish(text + ask, imgs, output_ids=torch.tensor(full[-1][-1]))

    def _img_gen_item(self, index):
        """rec_dataset.py:613-664 (train) / 666-717 (eval): history of (title, image code ids), query keywords, target = the
        next item's VQGAN code ids as ``img_k,`` tokens."""
        seq, imgs, text = self.seqs[index], [], ""
        codes = lambda item: "".join(f"img_{c}," for c in self.img_id2semantic[str(item)])
        for item in seq[-1 - self.history_len:-1]:
            imgs.append(self._image(item))
            np.random.random()                                 # extract_meta_gen draws (and ignores) p
            title = self.meta_data[str(item)]["title"]
            title = " ".join(("Unknown" if title == "" else title).split()[:20])
            text += f"<image> Title {title} ID {codes(item)} <|endofchunk|> "
        item = seq[-1]
        query = " ".join(self.meta_data[str(item)]["keywords"].split()[:30])
        if self.split == "train":
            return self._finish(text + f"Query: {query} What is the generated image ID to the query based on the history? <answer> {codes(item)}",
                                imgs, weight=1.0)
        return self._finish(text + f"Query: {query} What is the generated Image ID to the query based on the history? <answer>",
                            imgs, output_ids=codes(item), items=item)

    def __getitem__(self, index):
        if self.task == "img_gen":
            return self._img_gen_item(index)
        if self.task == "search":
            return self._search_item(index)
        if self.task == "exp":
            return self._exp_item(index)
        if self.task == "img_sel":
            return self._img_sel_item(index)
        seq = [it[0] for it in self.seqs[index]]
        imgs, text = [], ""
        if self.split == "train":                              # rec_dataset.py:372-456
            start = np.random.choice(list(range(0, len(seq) - self.history_len)), 1)[0]
            end = start + self.history_len
            for item in seq[start:end]:
                imgs.append(self._image(item))
                text += f"<image> {self.describe(item)} <answer> {self._item_token(item)} <|endofchunk|> "
            text += f"{self.QUESTION} <answer> {self._item_token(seq[end])}"
            ids, mask = self._tokenize(text)
            one = torch.LongTensor([1])
            net = {"input_ids": torch.cat([self.bos_item, ids, self.eos_item]), "attention_masks": torch.cat([one, mask, one]),
                   "patch_images": imgs if self.defer else torch.stack(imgs, dim=0), "weights": torch.tensor(2.0)}
            return {"net_input": net}
        test_len = 20 if self.subset == "hm" else 5            # rec_dataset.py:458-533
        for item in seq[-test_len:-1]:
            desc = self.describe(item)
            imgs.append(self._image(item))
            text += f"<image> {desc} {self._item_token(item)} <|endofchunk|> "
        text += f"{self.QUESTION} <answer>"
        ids, mask = self._tokenize(text)
        net = {"input_ids": ids, "attention_masks": mask, "patch_images": imgs if self.defer else torch.stack(imgs, dim=0),
               "input_len": len(text.split(" "))}
        return {"net_input": net, "net_output": {"output_ids": self._item_token(seq[-1])}}

    def collate(self, samples):
        if not self.defer:
            return collate_fn(samples, pad_idx=self.tokenizer.pad_token_id, eos_idx=self.tokenizer.eos_token_id)
        raw = [s["net_input"]["patch_images"] for s in samples]
        stub = [{"net_input": {**s["net_input"], "patch_images": torch.zeros(len(r), 0)}} for s, r in zip(samples, raw)]
        batch = collate_fn(stub, pad_idx=self.tokenizer.pad_token_id, eos_idx=self.tokenizer.eos_token_id)
        del batch["net_input"]["patch_images"]
        batch["net_input"]["patch_images_raw"] = raw           # [b][T] uint8 arrays -> ImagePreprocessor on the device
        return batch


class MixedRecDataset:
    """The task mixture of UniMP's multi-task training (rec_dataset.py:176-206, ``single_task`` off and ``task`` a list):
    every task but the LAST of the list contributes a random quarter of its users (``np.random.shuffle`` of the user keys at
    construction -- seed numpy before building it to reproduce a mixture), the last one all of them; samples carry their
    task's loss weight (2.0 rec, 1.0 others: cfg3's ``weights``)."""

    def __init__(self, folder, subset, tokenizer, tasks, **kw):
        self.parts, self.index = {}, []
        for i, t in enumerate(tasks):
            ds = self.parts[t] = RecDataset(folder, subset, tokenizer, split="train", task=t, **kw)
            keys = list(ds.data.keys())
            if i < len(tasks) - 1:
                np.random.shuffle(keys)
                keys = keys[:int(0.25 * len(keys))]
            ds.seqs = [ds.data[k] for k in keys]
            self.index += [(t, j) for j in range(len(ds.seqs))]
        self.tasks = [t for t, _ in self.index]
        self.collate = next(iter(self.parts.values())).collate

    def __len__(self):
        return len(self.index)

    def __getitem__(self, i):
        t, j = self.index[i]
        return self.parts[t][j]
