"""Does pulling the NEXT GEMM's weights into the memory-side cache on a second stream shorten a chain of weight-streaming decode GEMMs?
A chain of [N, K] GEMMs over rotating weights (1.2 GB: nothing survives in the 256 MB cache by itself), captured in one graph: plain, and with
unimp_prefetch of weights i + 1 forked beside GEMM i.  Not a pytest file."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimp_amd import ops, _lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 256
L = _lib.lib()
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
for N, K in [(10240, 2560), (2560, 10240), (7680, 2560), (2560, 2560)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    ws = [torch.randn(N, K, device="cuda").bfloat16() for _ in range(max(4, int(1.2e9 / (N * K * 2))))]
    for w in ws[:2]:
        ops.gemm(a, w)
    res = {}
    for mode in ("plain", "prefetch"):
        side = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        outs = []
        with torch.cuda.graph(g):
            main = torch.cuda.current_stream()
            for i, w in enumerate(ws):
                if mode == "prefetch" and i + 1 < len(ws):
                    side.wait_stream(main)                      # starts when GEMM i - 1 is done, i.e. beside GEMM i
                    nxt = ws[i + 1]
                    L.unimp_prefetch(nxt.data_ptr(), nxt.numel() * 2, blocks, sink.data_ptr(), side.cuda_stream)
                outs.append(ops.gemm(a, w))
            if mode == "prefetch":
                main.wait_stream(side)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); e1.synchronize()
        res[mode] = e0.elapsed_time(e1) * 1e3 / (5 * len(ws))
    print(f"M={M} N={N:6d} K={K:6d} plain {res['plain']:7.1f} us  with prefetch {res['prefetch']:7.1f} us   ({N * K * 2 / 1e6:.0f} MB; {N * K * 2 / res['prefetch'] / 1e6:.2f} TB/s)")
