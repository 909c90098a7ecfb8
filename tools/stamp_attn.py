"""Where the dK/dV kernel's time goes at a given shape: wave 0's s_memtime cycles per loop segment of every block (debug build).
Needs:  make -C unimp_amd/csrc EXTRA=-DATTN_STAMP OBJD=$PWD/build/obj_stamp OUT=$PWD/build/libunimp_hip_stamp.so
usage: stamp_attn.py [B H S D]   (causal, the LM's shape by default: 48 32 512 80)"""
import ctypes, os, sys, torch
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unimp_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "build", "libunimp_hip_stamp.so")          # the debug build, never the product library
from unimp_amd import ops
B, H, S, D = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (48, 32, 512, 80)
bf = torch.bfloat16
qkv = torch.randn(B, S, H, 3 * D, device="cuda").to(bf)
q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
dqkv = torch.empty_like(qkv)
dq, dk, dv = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
do = torch.randn(B, S, H, D, device="cuda").to(bf)
o, lse = ops.attn_fwd(q, k, v, D ** -0.5, ops.MASK_CAUSAL)
for _ in range(3):
    ops.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, D ** -0.5, ops.MASK_CAUSAL)
torch.cuda.synchronize()
buf = np.zeros(16384 * 12, dtype=np.uint64)
rc = _lib.lib().unimp_debug_attn_stamps(ctypes.c_void_p(buf.ctypes.data))
nb = min(16384, ((S + 63) // 64) * H * B)
t = buf.reshape(16384, 12)[:nb].astype(np.float64)
tiles = t[:, 9]
names = ["issue next tile's loads", "S, dP: fragment reads + MFMAs", "exp / dS / pack", "dV, dK: transposed reads + MFMAs", "wait loads + LDS stores", "barrier"]
loop = t[:, 7]
print(f"rc {rc}  B {B} H {H} S {S} D {D}: {nb} blocks stamped, {tiles.mean():.1f} query tiles per block; cycles per block (wave 0, s_memtime): "
      f"prologue {t[:, 6].mean():.0f}  loop {loop.mean():.0f}  epilogue {t[:, 8].mean():.0f};  wall per block {((t[:, 11] - t[:, 10]) / 100).mean():.2f} us")
per_tile = t[:, :6].sum(0) / tiles.sum()
for n, c in zip(names, per_tile):
    print(f"  {n:36s} {c:8.0f} cycles per tile  ({100 * c / per_tile.sum():4.1f} %)")
print(f"  {'sum':36s} {per_tile.sum():8.0f} cycles per tile; 22 MFMAs of 16 cycles = 352")
