"""The decoding loops (host logic) pinned against the installed transformers' own ``generate`` on a tiny GPT-NeoX:
same fp32 logits in, identical token sequences out (beam search incl. early stopping, num_return_sequences, EOS/pad)."""
import pytest
import torch

from unimp_amd.generate import beam_search, greedy_search


@pytest.fixture(scope="module")
def hf():
    from transformers import GPTNeoXConfig, GPTNeoXForCausalLM
    torch.manual_seed(0)
    c = GPTNeoXConfig(vocab_size=60, hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64,
                      max_position_embeddings=128, tie_word_embeddings=False)
    m = GPTNeoXForCausalLM(c).eval()
    for p in m.parameters():
        p.data.normal_(0, 0.6)             # peaky distributions: EOS gets picked within a few steps
    return m


def _logits_fn(m):
    return lambda seqs: m(input_ids=seqs).logits[:, -1]


@pytest.mark.parametrize("num_beams,nret,new", [(4, 4, 12), (10, 10, 20), (3, 1, 8)])
def test_beam_search_matches_transformers(hf, num_beams, nret, new):
    torch.manual_seed(1)
    for trial in range(4):
        ids = torch.randint(0, 60, (1, 7 + trial))
        eos = 5 + trial
        want = hf.generate(input_ids=ids, attention_mask=torch.ones_like(ids), num_beams=num_beams, num_return_sequences=nret,
                           early_stopping=True, max_new_tokens=new, eos_token_id=eos, pad_token_id=eos, do_sample=False,
                           length_penalty=1.0)
        got = beam_search(_logits_fn(hf), ids, num_beams, new, eos, eos, nret, early_stopping=True)
        assert got.shape == want.shape, (got.shape, want.shape)
        assert torch.equal(got, want), (trial, got.tolist(), want.tolist())


def test_greedy_matches_transformers(hf):
    ids = torch.randint(0, 60, (3, 9))
    want = hf.generate(input_ids=ids, attention_mask=torch.ones_like(ids), max_new_tokens=10, do_sample=False, eos_token_id=7,
                       pad_token_id=0)
    got = greedy_search(_logits_fn(hf), ids, 10, 7, 0)
    assert torch.equal(got, want)


def test_no_repeat_ngram(hf):
    ids = torch.randint(0, 60, (1, 6))
    want = hf.generate(input_ids=ids, attention_mask=torch.ones_like(ids), num_beams=3, max_new_tokens=12, do_sample=False,
                       eos_token_id=59, pad_token_id=59, no_repeat_ngram_size=2, early_stopping=True)
    got = beam_search(_logits_fn(hf), ids, 3, 12, 59, 59, 1, early_stopping=True, no_repeat_ngram_size=2)
    assert torch.equal(got, want)


def test_fused_beam_topk_is_a_gpu_path_only():
    """generate.beam_search takes unimp_beam_topk (log_softmax + beam scores + top-2K in two HIP launches) for GPU logits only: on CPU tensors the helper
    declines and the torch ops run -- the product path has no CPU kernel, and the CPU tests above pin the torch path against transformers."""
    import torch
    from unimp_amd import generate
    logits = torch.randn(4, 1000)
    assert generate._fused_topk(logits, torch.zeros(1, 4), 4) is None
    old = generate.BEAM_TOPK_FUSED
    try:
        generate.BEAM_TOPK_FUSED = False
        assert generate._fused_topk(logits, torch.zeros(1, 4), 4) is None
    finally:
        generate.BEAM_TOPK_FUSED = old
