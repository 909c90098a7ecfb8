"""ORACLE (test infrastructure only): optional emulation of the HIP path's STORAGE precision inside the fp32 oracle.

The product keeps activations in bf16 between kernels (fp32 inside them).  Every point where it writes a tensor to HBM is
tagged in the oracle modules with ``st(tag, x)`` -- the identity unless ``tag`` is switched on, in which case x is rounded to
bf16 and widened again (straight-through for autograd).  Tags:

  res      the residual stream after every sub-block (attention / MLP / gated cross-attention / gated FF / Perceiver steps)
  ln       LayerNorm / RMSNorm outputs (the GEMMs' A operands)
  gemm     GEMM outputs that are stored as such: fused QKV (after the rotary epilogue), to_q / to_kv, patch embedding
  act      the up-projection's activated output act(z) (z itself never leaves the registers)
  attn_p   the softmax probabilities as the P.V MFMA consumes them
  attn_o   attention outputs (the out-projection's A operand)
  vis      ViT tokens / Perceiver output handed to the language tower
  logits   the LM head's output

``with storage("res", "ln", ...):`` switches tags on for a block; ``ALL`` lists them.  Used by tests/error_budget.py (which
storage point contributes how much of the logits error) and by the parity tests' "same storage precision" reference."""
import contextlib
import torch

ALL = ("res", "ln", "gemm", "act", "attn_p", "attn_o", "vis", "logits")
_ON = set()


def st(tag, x):
    if tag in _ON and x.dtype == torch.float32:
        return x + (x.to(torch.bfloat16).to(torch.float32) - x).detach()
    return x


@contextlib.contextmanager
def storage(*tags):
    bad = [t for t in tags if t not in ALL]
    if bad:
        raise ValueError(f"unknown storage tags {bad}; known: {ALL}")
    old = set(_ON)
    _ON.clear()
    _ON.update(tags)
    try:
        yield
    finally:
        _ON.clear()
        _ON.update(old)
