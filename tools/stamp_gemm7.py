"""where a 256x256 tile of the one-wave-per-SIMD kernel (gemm7.hip, variant w4x) spends its time: per-block s_memtime / s_memrealtime
stamps around the prologue, the K loop and the epilogue.  Cycles per 64-k stage = K loop cycles / (K / 64 - ...): 2 048 is the matrix
pipe's floor (128 MFMAs x 16 cycles).  Needs the stamped debug build:
  make -C unimp_amd/csrc EXTRA=-DG7_STAMP OBJD=$PWD/build/obj_stamp7 OUT=$PWD/build/libunimp_hip_stamp7.so
usage: stamp_gemm7.py M N K [epilogue plain|bias|res|gelu2] [b_ks 0|1]"""
import ctypes, os, sys, torch
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unimp_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "build", "libunimp_hip_stamp7.so")          # the debug build, never the product library
from unimp_amd import ops
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (32768, 2560, 2560)))
epi = sys.argv[4] if len(sys.argv) > 4 else "plain"
bks = bool(int(sys.argv[5])) if len(sys.argv) > 5 else True
bf = torch.bfloat16
x = torch.randn(M, K, device="cuda").to(bf); w = (torch.randn((K, N) if bks else (N, K), device="cuda") * 0.05).to(bf)
out = torch.empty(M, N, dtype=bf, device="cuda")
kw = dict(out=out, variant="w4x", b_ks=bks)
if epi in ("bias", "res", "gelu2"):
    kw["bias"] = torch.randn(N, device="cuda").to(bf)
if epi == "res":
    kw["res"] = torch.randn(M, N, device="cuda").to(bf)
if epi == "gelu2":
    kw.update(act="gelu", pre=torch.empty((M, N), dtype=torch.uint8, device="cuda"), pre_deriv=True)
for _ in range(20):                      # the clock settles under load
    ops.gemm(x, w, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.gemm(x, w, **kw); e1.record(); torch.cuda.synchronize()
buf = np.zeros(8192 * 8, dtype=np.uint64)
rc = _lib.lib().unimp_debug_g7_stamps(ctypes.c_void_p(buf.ctypes.data))
nb = min(8192, ((M + 255) // 256) * ((N + 255) // 256))
t = buf.reshape(8192, 8)[:nb].astype(np.int64)
ms = e0.elapsed_time(e1)
cyc, rt = t[:, :4], t[:, 4:]
clock = (cyc[:, 2] - cyc[:, 1]) / np.maximum(1, rt[:, 2] - rt[:, 1]) * 100.0          # MHz in the K loop
stages = K // 64
loop = cyc[:, 2] - cyc[:, 1]
print(f"rc {rc}  w4x {epi} b_ks={int(bks)}  M,N,K = {M},{N},{K}  blocks {nb}  kernel {ms * 1e3:.1f} us = {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s")
print(f"in-kernel clock (K loop): median {np.median(clock):.0f} MHz (p10 {np.percentile(clock, 10):.0f}, p90 {np.percentile(clock, 90):.0f})")
print(f"per block (cycles): prologue {np.median(cyc[:, 1] - cyc[:, 0]):.0f}  K loop {np.median(loop):.0f} = {np.median(loop) / stages:.0f} per 64-k stage (floor 2048; p10 {np.percentile(loop, 10) / stages:.0f} p90 {np.percentile(loop, 90) / stages:.0f})  "
      f"epilogue {np.median(cyc[:, 3] - cyc[:, 2]):.0f}")
print(f"per block (us, 100 MHz counter): prologue {np.median(rt[:, 1] - rt[:, 0]) / 100:.2f}  K loop {np.median(rt[:, 2] - rt[:, 1]) / 100:.2f}  epilogue {np.median(rt[:, 3] - rt[:, 2]) / 100:.2f}  "
      f"first entry -> last exit {(rt[:, 3].max() - rt[:, 0].min()) / 100:.1f}")
