"""Do the GEMM variants a TINY-model step takes at 64 < M < 512 give the same bits launch after launch (and the same as v1)?  A second process keeps the
GPU busy meanwhile (the eight-rank test shares one GPU among eight processes).  Not a pytest file."""
import os, sys, subprocess, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimp_amd import ops

if len(sys.argv) > 1 and sys.argv[1] == "noise":
    a = torch.randn(4096, 4096, device="cuda").bfloat16()
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[2]):
        for _ in range(20):
            ops.gemm(a, a)
        torch.cuda.synchronize()
    sys.exit(0)
noise = [subprocess.Popen([sys.executable, __file__, "noise", "40"]) for _ in range(3)]
time.sleep(8)
g = torch.Generator().manual_seed(3)
bad = 0
for M, N, K in [(128, 128, 512), (96, 384, 256), (160, 512, 256), (128, 128, 256), (200, 128, 384), (72, 640, 512), (510, 256, 1024), (128, 384, 128)]:
    a = torch.randn(M, K, generator=g).bfloat16().cuda(); b = torch.randn(N, K, generator=g).bfloat16().cuda()
    bias = torch.randn(N, generator=g).bfloat16().cuda(); res = torch.randn(M, N, generator=g).bfloat16().cuda()
    for kw in (dict(), dict(bias=bias, act="gelu"), dict(bias=bias, res=res)):
        base1 = ops.gemm(a, b, variant=1, **kw)
        auto = ops.gemm(a, b, **kw)
        same_v = torch.equal(base1, auto)
        nd = 0
        for _ in range(400):
            nd += int(not torch.equal(auto, ops.gemm(a, b, **kw)))
        if nd or not same_v:
            bad += 1
        print(f"M={M} N={N} K={K} {sorted(kw)}: auto == v1 {same_v}; launches that differ from the first: {nd} of 400")
for p in noise:
    p.wait()
print("shapes with a difference:", bad)
