"""ad-hoc: where a 256x256 tile's wall time goes (needs a build with EXTRA=-DG3_STAMP): per-block s_memrealtime stamps grouped by CU."""
import ctypes, os, sys, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops, _lib
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (98688, 1024, 1024)))
x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    ops.gemm(x, w, out=out, variant="pp256")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.gemm(x, w, out=out, variant="pp256"); e1.record(); torch.cuda.synchronize()
buf = np.zeros(8192 * 12, dtype=np.uint64)
rc = _lib.lib().unimp_debug_g3_stamps(ctypes.c_void_p(buf.ctypes.data))
nb = min(8192, ((M + 255) // 256) * ((N + 255) // 256))
t = buf.reshape(8192, 12)[:nb].astype(np.int64)
print(f"rc {rc}  M,N,K = {M},{N},{K}  blocks {nb}  kernel {e0.elapsed_time(e1) * 1e3:.1f} us (event)  first entry -> last exit {(t[:, 3].max() - t[:, 0].min()) / 100:.1f} us")
pro, loop, epi = (t[:, 1] - t[:, 0]) / 100.0, (t[:, 2] - t[:, 1]) / 100.0, (t[:, 3] - t[:, 2]) / 100.0
print(f"per block (us): prologue {pro.mean():.2f} (p90 {np.percentile(pro, 90):.2f})  K loop {loop.mean():.2f}  epilogue {epi.mean():.2f} (p90 {np.percentile(epi, 90):.2f})  total {(pro + loop + epi).mean():.2f}")
d = lambda a, b: ((t[:, a] - t[:, b]) / 100.0).mean()
print(f"epilogue pieces (us): fetch {d(4, 2):.2f}  stage0 {d(5, 4):.2f}  pass0 {d(6, 5):.2f}  stage1 {d(7, 6):.2f}  pass1 {d(3, 7):.2f}")
sys.exit(0)
key = t[:, 10] * 1000 + (t[:, 11] & 0xf)          # HW_ID + XCC_ID
gaps, per_cu = [], {}
for i in range(nb):
    per_cu.setdefault(int(key[i]), []).append(i)
for k, idx in per_cu.items():
    idx.sort(key=lambda i: t[i, 0])
    for a, b in zip(idx[:-1], idx[1:]):
        gaps.append((t[b, 0] - t[a, 3]) / 100.0)
gaps = np.array(gaps)
print(f"distinct (HW_ID, XCC) keys {len(per_cu)}; blocks per key {nb / len(per_cu):.1f}; gap exit(prev, wave 0) -> entry(next) on the same key: mean {gaps.mean():.2f} us, median {np.median(gaps):.2f}, p90 {np.percentile(gaps, 90):.2f}")
