"""``create_model_and_transforms`` with open_flamingo's signature (call sites UniMP/mmrec.py:476-524).

No network exists on the build or GPU boxes, so pretrained weights are never downloaded here: the towers are
built from the named architecture (random init) and weights come from ``load_state_dict(strict=False)`` exactly
as UniMP does with ``checkpoint.pt`` (mmrec.py:513-514).  The tokenizer is the HF one when it can be loaded from
local files, else the built-in ``SyntheticTokenizer`` (same special-token protocol).
"""
import re
import warnings
import torch
import torch.nn as nn

from .flamingo import Flamingo, freeze_like_factory
from .lm import build_lm
from .vit import VisionTransformer, CLIPStub, VISION_CONFIGS

FLAMINGO_MEAN = [0.48145466, 0.4578275, 0.40821073]      # rec_dataset.py:30-31 (unify_dataset.py:34 rounds them to 3 digits;
FLAMINGO_STD = [0.26862954, 0.26130258, 0.27577711]       #  the recommendation pipeline this build targets uses the full ones)


class SyntheticTokenizer:
    """Minimal HF-tokenizer lookalike: whitespace / punctuation pieces hashed into a fixed base vocabulary, plus
    added special tokens that are never split (what mmrec.py:538-581 relies on)."""

    def __init__(self, base_vocab=50277, bos_token="<|endoftext|>", eos_token="<|endoftext|>"):
        self.base_vocab = base_vocab
        self.added = {}
        self.bos_token, self.eos_token, self.pad_token = bos_token, eos_token, None
        self.bos_token_id = self.eos_token_id = 0
        self.pad_token_id = None
        self.model_max_length = 2048
        self.padding_side = "right"

    def __len__(self):
        return self.base_vocab + len(self.added)

    def add_special_tokens(self, d):
        n = 0
        for k, v in d.items():
            toks = v if isinstance(v, (list, tuple)) else [v]
            for t in toks:
                if t not in self.added:
                    self.added[t] = self.base_vocab + len(self.added)
                    n += 1
            if k == "pad_token":
                self.pad_token, self.pad_token_id = v, self.added[v]
        self._rx = None
        return n

    def add_tokens(self, toks):
        return self.add_special_tokens({"additional_special_tokens": list(toks)})

    def _regex(self):
        if getattr(self, "_rx", None) is None:
            alts = sorted(self.added, key=len, reverse=True)
            self._rx = re.compile("(" + "|".join(map(re.escape, alts)) + r")|(\w+|[^\w\s])") if alts else re.compile(r"()(\w+|[^\w\s])")
        return self._rx

    def encode(self, text, add_special_tokens=False):
        ids = []
        for m in self._regex().finditer(text):
            if m.group(1):
                ids.append(self.added[m.group(1)])
            else:
                h = 0
                for ch in m.group(2):
                    h = (h * 131 + ord(ch)) % 2147483647
                ids.append(1 + h % (self.base_vocab - 1))
        return ids

    def __call__(self, text, add_special_tokens=False, return_tensors=None, truncation=False, max_length=None, **kw):
        single = isinstance(text, str)
        seqs = [self.encode(t) for t in ([text] if single else text)]
        if truncation:
            seqs = [s[:(max_length or self.model_max_length)] for s in seqs]
        if return_tensors == "pt":
            L = max(len(s) for s in seqs)
            pad = self.pad_token_id if self.pad_token_id is not None else 0
            ids = torch.tensor([s + [pad] * (L - len(s)) for s in seqs])
            mask = torch.tensor([[1] * len(s) + [0] * (L - len(s)) for s in seqs])
            return {"input_ids": ids, "attention_mask": mask}
        return {"input_ids": seqs[0] if single else seqs, "attention_mask": [1] * len(seqs[0]) if single else [[1] * len(s) for s in seqs]}

    def batch_decode(self, seqs, skip_special_tokens=False):
        return [self.decode(s, skip_special_tokens) for s in seqs]

    def decode(self, ids, skip_special_tokens=False):
        inv = {v: k for k, v in self.added.items()}
        return " ".join(inv.get(int(i), f"tok{int(i)}") for i in ids if not (skip_special_tokens and int(i) in inv))


class ImageProcessor:
    """PIL image -> normalised (3, S, S) fp32 tensor: bicubic resize + ToTensor + Normalize (rec_dataset.py:90-107)."""

    def __init__(self, size=224):
        self.size = size
        self.mean = torch.tensor(FLAMINGO_MEAN).view(3, 1, 1)
        self.std = torch.tensor(FLAMINGO_STD).view(3, 1, 1)

    def __call__(self, img):
        import numpy as np
        from PIL import Image
        img = img.convert("RGB").resize((self.size, self.size), Image.BICUBIC)
        x = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)
        return (x - self.mean) / self.std

    def batch(self, images, device="cuda", dtype=torch.bfloat16):
        """the same transform for a list of decoded images on the GPU (unimp_amd.data.ImagePreprocessor: HIP kernels,
        bit-exact with Pillow's resize) -> [n, 3, S, S] on `device`."""
        from .data import ImagePreprocessor
        key = (str(device), dtype)
        if getattr(self, "_dev", None) is None or self._dev[0] != key:
            self._dev = (key, ImagePreprocessor(self.size, device=device, dtype=dtype))
        return self._dev[1](images)


def _load_tokenizer(tokenizer_path, use_local_files):
    try:
        from transformers import AutoTokenizer
        return AutoTokenizer.from_pretrained(tokenizer_path, local_files_only=True)
    except Exception as e:  # no network / not cached
        warnings.warn(f"tokenizer {tokenizer_path!r} not available locally ({type(e).__name__}); using SyntheticTokenizer")
        return SyntheticTokenizer()


def create_model_and_transforms(clip_vision_encoder_path, clip_vision_encoder_pretrained, lang_encoder_path, tokenizer_path,
                                cross_attn_every_n_layers=1, use_local_files=False, decoder_layers_attr_name=None,
                                freeze_lm_embeddings=False, device=None, dtype=torch.bfloat16, tokenizer=None,
                                **flamingo_kwargs):
    """Returns (model, image_processor, tokenizer) like open_flamingo's factory.

    ``clip_vision_encoder_path`` / ``lang_encoder_path`` may be names ("ViT-L-14", ".../RedPajama-INCITE-Instruct-3B-v1",
    "facebook/opt-125m") or explicit config objects / dicts (tests use tiny ones)."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    with torch.device(device):          # build directly in device memory (4B params: no 17 GB host staging)
        return _build(clip_vision_encoder_path, lang_encoder_path, tokenizer_path, cross_attn_every_n_layers, use_local_files,
                      decoder_layers_attr_name, freeze_lm_embeddings, dtype, tokenizer, flamingo_kwargs)


def _build(clip_vision_encoder_path, lang_encoder_path, tokenizer_path, cross_attn_every_n_layers, use_local_files,
           decoder_layers_attr_name, freeze_lm_embeddings, dtype, tokenizer, flamingo_kwargs):
    vcfg = clip_vision_encoder_path if isinstance(clip_vision_encoder_path, dict) else VISION_CONFIGS[clip_vision_encoder_path]
    visual = VisionTransformer(**vcfg)
    visual.output_tokens = True
    image_processor = ImageProcessor(vcfg.get("image_size", 224))

    text_tokenizer = tokenizer or _load_tokenizer(tokenizer_path, use_local_files)
    text_tokenizer.add_special_tokens({"additional_special_tokens": ["<|endofchunk|>", "<image>"]})
    if text_tokenizer.pad_token is None:
        text_tokenizer.add_special_tokens({"pad_token": "<PAD>"})

    lang_encoder = build_lm(lang_encoder_path)
    if decoder_layers_attr_name is not None:
        lang_encoder.decoder_layers_attr = decoder_layers_attr_name
    lang_encoder.resize_token_embeddings(len(text_tokenizer))

    model = Flamingo(CLIPStub(visual), lang_encoder,
                     text_tokenizer.encode("<|endofchunk|>")[-1], text_tokenizer.encode("<image>")[-1],
                     vis_dim=vcfg["width"], cross_attn_every_n_layers=cross_attn_every_n_layers, **flamingo_kwargs)
    freeze_like_factory(model, freeze_lm_embeddings)
    model.to(dtype=dtype)
    return model, image_processor, text_tokenizer
