"""ad-hoc: per-half-stage cycle stamps of the one-wave-per-SIMD GEMM (needs a build with EXTRA=-DG4_STAMP)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops, _lib
import numpy as np
M = N = K = 8192
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    ops.gemm(a, b, out=out, variant="w4")
torch.cuda.synchronize()
buf = np.zeros(4 * 64 * 4, dtype=np.uint64)
lib = _lib.lib() if hasattr(_lib, "lib") else _lib.LIB
rc = lib.unimp_debug_g4_stamps(ctypes.c_void_p(buf.ctypes.data))
t = buf.reshape(4, 64, 4).astype(np.int64)
print("rc", rc)
for w in range(4):
    print(f"wave {w}:  h  vmcnt-wait  barrier  body(rows)  lgkm-wait  total")
    for h in range(8, 40):
        nxt = t[w, h + 1, 0]
        print(f"   {h:3d} {t[w,h,1]-t[w,h,0]:8d} {t[w,h,2]-t[w,h,1]:8d} {t[w,h,3]-t[w,h,2]:8d} {nxt-t[w,h,3]:8d} {nxt-t[w,h,0]:8d}")
    d = t[w, 9:41, 0] - t[w, 8:40, 0]
    print("   mean total", d.mean(), " mean body", (t[w, 8:40, 3] - t[w, 8:40, 2]).mean(), " barrier", (t[w, 8:40, 2] - t[w, 8:40, 1]).mean(), " vmwait", (t[w, 8:40, 1] - t[w, 8:40, 0]).mean())
