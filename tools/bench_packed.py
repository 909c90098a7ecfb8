"""packed-B ping-pong GEMM vs the unpacked kernels at the frozen-tower shapes (b = 64)."""
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
M = int(os.environ.get("M", 32768))
for (N, K, b_ks, epi) in [(10240, 2560, 0, 0), (2560, 10240, 0, 1), (10240, 2560, 1, 1), (2560, 10240, 1, 0), (7680, 2560, 0, 0), (2560, 7680, 1, 0), (2560, 2560, 0, 1), (2560, 2560, 1, 0)]:
    x = torch.randn(M, K, device="cuda").to(bf)
    w = torch.randn((K, N) if b_ks else (N, K), device="cuda").to(bf)
    res = torch.randn(M, N, device="cuda").to(bf) if epi else None
    out = torch.empty(M, N, dtype=bf, device="cuda")
    pk = ops.pack_b(w, bool(b_ks))
    fl = 2.0 * M * N * K
    r = {}
    for var in ("pp256", "w8", "pp256p", "pk256", "pk128"):
        try:
            t = timeit(lambda: ops.gemm(x, w, b_ks=bool(b_ks), res=res, out=out, variant=var, b_pk=pk if var.startswith("pk") else None))
            r[var] = fl / t / 1e9
        except Exception as e:
            r[var] = float("nan")
    print(f"M={M} N={N:6d} K={K:6d} b_ks={b_ks} res={epi}: " + "  ".join(f"{k} {v:6.0f}" for k, v in r.items()), flush=True)
