"""MX-fp8 GEMM (unimp_gemm_mxfp8) vs the bf16 GEMM at the frozen-tower shapes: TFLOP/s and fraction of the MX-fp8 dense peak (5 PF)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
bf = torch.bfloat16
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
for (M, N, K) in [(8192, 8192, 8192), (12288, 16384, 4096), (12288, 4096, 16384), (12288, 12288, 4096), (12288, 4096, 4096), (24576, 10240, 2560), (24576, 2560, 10240)]:
    x = torch.randn(M, K, device="cuda").to(bf); w = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
    out = torch.empty(M, N, dtype=bf, device="cuda")
    tq = timeit(lambda: ops.mx_quantize(x))
    a, b = ops.mx_quantize(x), ops.mx_quantize(w)
    tm = timeit(lambda: ops.gemm_mx(a, b, out=out))
    tb = timeit(lambda: ops.gemm(x, w, out=out))
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:6d} K={K:6d}: mxfp8 {tm:.3f} ms = {fl / tm / 1e9:.0f} TF ({fl / tm / 1e9 / 5000:.1%} of 5 PF) | quantise x {tq * 1e3:.0f} us "
          f"({M * K * 3.03 / tq / 1e9:.2f} TB/s) | bf16 {tb:.3f} ms = {fl / tb / 1e9:.0f} TF", flush=True)
