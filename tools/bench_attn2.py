"""A/B of the attention kernel generations at the step's shapes (b = 48; lm64: b = 64): ms and TFLOP/s (4 S_q S_k D per (b, h), full
count).  Generation 2 = the default (second-generation forward / dQ; dK/dV: attention3.hip where it serves the form, else first
generation), 3 = second generation throughout, 4 = generation 2 without attention3.hip (the round-4 default)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops, _lib
torch.manual_seed(0)
bf = torch.bfloat16
SHAPES = {"lm": (48, 32, 512, 512, 80, 1), "lm64": (64, 32, 512, 512, 80, 1), "lm3": (3, 32, 512, 512, 80, 1), "lm6": (6, 32, 512, 512, 80, 1), "lm8": (8, 32, 512, 512, 80, 1),
          "lm16": (16, 32, 512, 512, 80, 1), "lm32": (32, 32, 512, 512, 80, 1), "lm1k": (16, 32, 1024, 1024, 80, 1), "vit": (384, 16, 257, 257, 64, 0), "xattn": (48, 8, 512, 512, 64, 2),
          "perc": (384, 8, 64, 320, 64, 0), "mpt": (8, 32, 1024, 1024, 128, 1), "lm2k": (4, 32, 2048, 2048, 80, 1)}


def setup(B, H, Sq, Sk, D, mode):
    if Sq == Sk:
        qkv = torch.randn(B, Sq, H, 3 * D, device="cuda").to(bf)
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
    else:
        q = torch.randn(B, Sq, H, D, device="cuda").to(bf)
        kv = torch.randn(B, Sk, 2, H, D, device="cuda").to(bf)
        k, v = kv[:, :, 0], kv[:, :, 1]
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        dk, dv = dkv[:, :, 0], dkv[:, :, 1]
    seg, seg_len, kv_len = None, 0, None
    if mode == 2:
        seg_len = 64
        seg = (torch.arange(Sq, device="cuda", dtype=torch.int32) * (Sk // 64) // Sq + 1).clamp(max=Sk // 64)[None].expand(B, -1).contiguous()
    if mode == 1:
        kv_len = torch.randint(int(0.75 * Sk), Sk + 1, (B,), device="cuda", dtype=torch.int32)
    do = torch.randn(B, Sq, H, D, device="cuda").to(bf)
    return q, k, v, dq, dk, dv, do, kv_len, seg, seg_len


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


for name in (sys.argv[1:] or SHAPES):
    B, H, Sq, Sk, D, mode = SHAPES[name]
    q, k, v, dq, dk, dv, do, kv_len, seg, seg_len = setup(B, H, Sq, Sk, D, mode)
    fl = 4.0 * B * H * Sq * Sk * D
    res = {}
    for gen in (1, 2, 3, 4):
        _lib.lib().unimp_attn_set_generation(gen)
        o, lse = ops.attn_fwd(q, k, v, D ** -0.5, mode, kv_len, seg, seg_len)
        tf = timeit(lambda: ops.attn_fwd(q, k, v, D ** -0.5, mode, kv_len, seg, seg_len))
        tb = timeit(lambda: ops.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, D ** -0.5, mode, kv_len, seg, seg_len))
        res[gen] = (tf, tb, o.float().clone(), dq.float().clone(), dk.float().clone(), dv.float().clone())
    d = [float((res[1][i] - res[2][i]).abs().max()) for i in range(2, 6)]
    d4 = [float((res[4][i] - res[2][i]).abs().max()) for i in range(4, 6)]
    print(f"{name:6s} B{B} H{H} {Sq}x{Sk} D{D} mode{mode}: fwd gen1 {res[1][0]:.3f} ms ({fl / res[1][0] / 1e9:.0f} TF) gen2 {res[2][0]:.3f} ms ({fl / res[2][0] / 1e9:.0f} TF) | "
          f"bwd gen1 {res[1][1]:.3f} ms ({2.5 * fl / res[1][1] / 1e9:.0f} TF) gen2 {res[2][1]:.3f} ms gen3 {res[3][1]:.3f} ms gen4 {res[4][1]:.3f} ms | max|gen4-gen2| dk {d4[0]:.2e} dv {d4[1]:.2e} | max|gen1-gen2| o {d[0]:.2e} dq {d[1]:.2e} dk {d[2]:.2e} dv {d[3]:.2e}", flush=True)
