import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from unimp_amd import ops
names = {v: k for k, v in ops.GEMM_VARIANTS.items()}
for M in (469, 256, 128):
    for N, K in [(10240, 2560), (2560, 10240), (7680, 2560), (2560, 2560)]:
        a = torch.randn(M, K, device="cuda").bfloat16()
        ws = [torch.randn(N, K, device="cuda").bfloat16() for _ in range(max(2, int(0.6e9 / (N * K * 2))))]
        res = []
        for v in (1, 4, 5, 10, 11, 13, 14, 18, 19, 15):
            try:
                for w in ws[:2]: ops.gemm(a, w, variant=v)
            except Exception as e:
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                for w in ws: ops.gemm(a, w, variant=v)
            e1.record(); e1.synchronize()
            res.append((e0.elapsed_time(e1) * 1e3 / (3 * len(ws)), names.get(v, v)))
        res.sort()
        print(f"M={M} N={N} K={K}: " + "  ".join(f"{n} {t:.0f}us" for t, n in res[:6]) + f"   | v1 {[t for t, n in res if n == 'v1'][0]:.0f}us")
