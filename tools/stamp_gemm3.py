"""where a 256x256 tile's wall time goes: per-block s_memrealtime stamps (100 MHz) of the ping-pong kernels, grouped by CU.
Needs the stamped debug build:  make -C unimp_amd/csrc EXTRA=-DG3_STAMP OBJD=$PWD/build/obj_stamp OUT=$PWD/build/libunimp_hip_stamp.so
usage: stamp_gemm3.py M N K [variant pp256|pp256x] [epilogue plain|bias|res|aux|gelu2] [b_ks 0|1]"""
import ctypes, os, sys, torch
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unimp_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "build", "libunimp_hip_stamp.so")          # the debug build, never the product library
from unimp_amd import ops
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (98688, 1024, 1024)))
variant = sys.argv[4] if len(sys.argv) > 4 else "pp256"
epi = sys.argv[5] if len(sys.argv) > 5 else "plain"
bks = bool(int(sys.argv[6])) if len(sys.argv) > 6 else False
bf = torch.bfloat16
x = torch.randn(M, K, device="cuda").to(bf); w = (torch.randn((K, N) if bks else (N, K), device="cuda") * 0.05).to(bf)
out = torch.empty(M, N, dtype=bf, device="cuda")
kw = dict(out=out, variant=variant, b_ks=bks)
if epi in ("bias", "res", "gelu2"):
    kw["bias"] = torch.randn(N, device="cuda").to(bf)
if epi == "res":
    kw["res"] = torch.randn(M, N, device="cuda").to(bf)
if epi == "aux":
    kw.update(aux=torch.randint(0, 256, (M, N), dtype=torch.uint8, device="cuda"), dact="deriv")
if epi == "gelu2":
    kw.update(act="gelu", pre=torch.empty((M, N), dtype=torch.uint8, device="cuda"), pre_deriv=True)
for _ in range(3):
    ops.gemm(x, w, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.gemm(x, w, **kw); e1.record(); torch.cuda.synchronize()
buf = np.zeros(8192 * 12, dtype=np.uint64)
fn = getattr(_lib.lib(), {"x": "unimp_debug_g3x_stamps", "a": "unimp_debug_g3a_stamps", "b": "unimp_debug_g3b_stamps"}.get(variant[-1], "unimp_debug_g3_stamps"))
rc = fn(ctypes.c_void_p(buf.ctypes.data))
nb = min(8192, ((M + 255) // 256) * ((N + 255) // 256))
t = buf.reshape(8192, 12)[:nb].astype(np.int64)
ms = e0.elapsed_time(e1)
print(f"rc {rc}  {variant} {epi} b_ks={int(bks)}  M,N,K = {M},{N},{K}  blocks {nb}  kernel {ms * 1e3:.1f} us (event) = {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s  "
      f"first entry -> last exit {(t[:, 3].max() - t[:, 0].min()) / 100:.1f} us")
pro, loop, epi_t = (t[:, 1] - t[:, 0]) / 100.0, (t[:, 2] - t[:, 1]) / 100.0, (t[:, 3] - t[:, 2]) / 100.0
print(f"per block (us): prologue {pro.mean():.2f} (p90 {np.percentile(pro, 90):.2f})  K loop {loop.mean():.2f} (p10 {np.percentile(loop, 10):.2f} p90 {np.percentile(loop, 90):.2f})  "
      f"epilogue {epi_t.mean():.2f} (p90 {np.percentile(epi_t, 90):.2f})  total {(pro + loop + epi_t).mean():.2f}")
d = lambda a, b: ((t[:, a] - t[:, b]) / 100.0).mean()
print(f"epilogue pieces (us): fetch {d(4, 2):.2f}  stage0 {d(5, 4):.2f}  pass0 {d(6, 5):.2f}  stage1 {d(7, 6):.2f}  pass1 {d(3, 7):.2f}")
key = t[:, 10] * 1000 + (t[:, 11] & 0xf)          # HW_ID + XCC_ID
gaps, per_cu = [], {}
for i in range(nb):
    per_cu.setdefault(int(key[i]), []).append(i)
for k, idx in per_cu.items():
    idx.sort(key=lambda i: t[i, 0])
    for a, b in zip(idx[:-1], idx[1:]):
        gaps.append((t[b, 0] - t[a, 3]) / 100.0)
gaps = np.array(gaps) if gaps else np.zeros(1)
rounds = nb / max(1, len(per_cu))
span = (t[:, 3].max() - t[:, 0].min()) / 100.0
print(f"distinct (HW_ID, XCC) keys {len(per_cu)}; blocks per key {rounds:.2f}; gap exit(prev, wave 0) -> entry(next) on the same key: mean {gaps.mean():.2f} us, "
      f"median {np.median(gaps):.2f}, p90 {np.percentile(gaps, 90):.2f};  span / rounds = {span / max(rounds, 1):.2f} us per tile slot")
