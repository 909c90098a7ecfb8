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

  grad     (round 6; not part of ALL) the BACKWARD side of every tag that is on: the gradient flowing back through a storage point is
           rounded to bf16 as well -- the product keeps dX tensors (d residual, d LayerNorm output, dqkv, d act ...) in bf16 between its
           backward kernels exactly where it keeps the activations in bf16 on the way forward.  ``ALL_BWD = ALL + ("grad",)`` is the
           yardstick of the full-depth GRADIENT parity (bench.full_depth_parity: parity.gradients); parameter gradients themselves are
           rounded to bf16 by the caller (the product's flat gradient buffer is bf16).

``with storage("res", "ln", ...):`` switches tags on for a block; ``ALL`` lists them.  Used by tests/error_budget.py (which
storage point contributes how much of the logits error) and by the parity tests' "same storage precision" reference."""
import contextlib
import torch

ALL = ("res", "ln", "gemm", "act", "attn_p", "attn_o", "vis", "logits")
ALL_BWD = ALL + ("grad",)
_ON = set()


class _StoreBoth(torch.autograd.Function):
    """bf16 storage point in both directions: the value on the way forward, its gradient on the way back"""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def st(tag, x):
    if tag in _ON and x.dtype == torch.float32:
        if "grad" in _ON and x.requires_grad:
            return _StoreBoth.apply(x)
        return x + (x.to(torch.bfloat16).to(torch.float32) - x).detach()
    return x


@contextlib.contextmanager
def storage(*tags):
    bad = [t for t in tags if t not in ALL_BWD]
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


# ---- MX-fp8 emulation of the frozen towers' GEMMs (cfg5: "fp8 MFMA weights"; the product's path is csrc/mx.hip) ----------------
def mx_dequant(x):
    """x [..., K] -> the values an MX-fp8 operand holds: per 32 consecutive k one E8M0 scale 2^(floor(log2 amax) - 8), elements
    x / scale saturated to +-448 and rounded to OCP e4m3 (round to nearest even), widened again."""
    K = x.shape[-1]
    assert K % 32 == 0
    xb = x.float().reshape(-1, K // 32, 32)
    amax = xb.abs().amax(-1, keepdim=True)
    e = torch.where(amax > 0, torch.floor(torch.log2(amax.clamp_min(1e-38))), torch.full_like(amax, -127.0))
    scale = torch.pow(2.0, (e - 8).clamp(-127, 127))
    q = (xb / scale).clamp(-448, 448).to(torch.float8_e4m3fn).float() * scale
    return q.reshape(x.shape)


class _MXLinear(torch.autograd.Function):
    """y = Q(x) Q(W)^T + b with both operands quantised along the contraction (in); dx = Q(dy) Q'(W) with dy and W^T quantised
    along `out` -- the same e4m3 weights seen along the other dimension, as functional._frozen_mx(w, transposed=True) does."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(w)
        y = mx_dequant(x) @ mx_dequant(w).t()
        return y if b is None else y + b

    @staticmethod
    def backward(ctx, dy):
        (w,) = ctx.saved_tensors
        wt = mx_dequant(w.t().contiguous())               # [in, out], blocks along out
        return mx_dequant(dy.contiguous()) @ wt.t(), None, None


@contextlib.contextmanager
def mx_frozen(min_rows=65, min_dim=2048):
    """inside: every F.linear on a FROZEN weight (requires_grad False) whose two dimensions are multiples of 128 and at least
    ``min_dim`` (the product's FP8_MIN_DIM: the ViT-L/14's 1024-wide projections stay bf16), applied to more than 64 rows, runs as an
    emulated MX-fp8 product (functional._mx_ok's rule); everything else stays fp32."""
    import torch.nn.functional as F
    real = F.linear

    def linear(x, w, b=None):
        rows = x.numel() // x.shape[-1]
        if (not w.requires_grad) and w.dim() == 2 and w.shape[0] % 128 == 0 and w.shape[1] % 128 == 0 and rows >= min_rows and min(w.shape) >= min_dim:
            return _MXLinear.apply(x, w, b)
        return real(x, w, b)
    F.linear = linear
    try:
        yield
    finally:
        F.linear = real
