"""Interleaved A/B of GEMM kernel variants on the step's shapes and epilogue forms, ONE process (cdna guide rule 24): every round
times every variant once, rounds alternate, the table reports the median TFLOP/s per (shape, epilogue, variant) and whether each
variant's output bits equal the first variant's.  usage: bench_gemm_ab.py [rounds] [variant,variant,...] [shape-filter]"""
import sys, os, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
bf = torch.bfloat16
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
VARIANTS = sys.argv[2].split(",") if len(sys.argv) > 2 else ["pp256", "pp256x", "pp256p", "pp256px", "w8"]
FILT = sys.argv[3] if len(sys.argv) > 3 else ""
B = int(os.environ.get("AB_BATCH", "64"))
ML, MV = B * 512, B * 8 * 257
# (name, M, N, K, a_ks, b_ks, epilogue)   epilogue: plain | bias | gelu2 (GELU + stored GELU' u8) | aux (x stored u8 derivative) | res (bias + residual) | qgelu | rope
SHAPES = [
    ("lm up  +gelu+gelu'", ML, 10240, 2560, 0, 0, "gelu2"),
    ("lm dX(up) x aux   ", ML, 2560, 10240, 0, 1, "plain"),
    ("lm down +res      ", ML, 2560, 10240, 0, 0, "res"),
    ("lm dX(down) x aux ", ML, 10240, 2560, 0, 1, "aux"),
    ("lm qkv +rope      ", ML, 7680, 2560, 0, 0, "rope"),
    ("lm dX(qkv)        ", ML, 2560, 7680, 0, 1, "plain"),
    ("vit-like qkv +rope", MV, 3072, 1024, 0, 0, "rope64"),
    ("lm attn-out +res  ", ML, 2560, 2560, 0, 0, "res"),
    ("lm dX(attn-out)   ", ML, 2560, 2560, 0, 1, "plain"),
    ("vit up +qgelu     ", MV, 4096, 1024, 0, 0, "qgelu"),
    ("vit down +res     ", MV, 1024, 4096, 0, 0, "res"),
    ("vit qkv +bias     ", MV, 3072, 1024, 0, 0, "bias"),
    ("vit out +res      ", MV, 1024, 1024, 0, 0, "res"),
    # forward GEMMs with the (frozen) weight read from a transposed copy: B k-strided, whole 512-B lines per LDS-DMA row
    ("lm up  W^T +gelu2 ", ML, 10240, 2560, 0, 1, "gelu2"),
    ("lm down W^T +res  ", ML, 2560, 10240, 0, 1, "res"),
    ("lm qkv W^T +bias  ", ML, 7680, 2560, 0, 1, "bias"),
    ("lm qkv     +bias  ", ML, 7680, 2560, 0, 0, "bias"),
    ("lm attn-out W^T+res", ML, 2560, 2560, 0, 1, "res"),
    ("vit up W^T +qgelu ", MV, 4096, 1024, 0, 1, "qgelu"),
    ("vit down W^T +res ", MV, 1024, 4096, 0, 1, "res"),
    ("vit qkv W^T +bias ", MV, 3072, 1024, 0, 1, "bias"),
    ("vit out W^T +res  ", MV, 1024, 1024, 0, 1, "res"),
    ("xattn dW          ", 10240, 2560, ML, 1, 1, "plain"),
    ("xattn dW^T        ", 2560, 10240, ML, 1, 1, "plain"),
]


def make(M, N, K, aks, bks, epi):
    g = torch.Generator(device="cuda"); g.manual_seed(M + N + K)
    a = torch.randn((K, M) if aks else (M, K), device="cuda", generator=g).to(bf)
    b = (torch.randn((K, N) if bks else (N, K), device="cuda", generator=g) * 0.05).to(bf)
    out = torch.empty((M, N), dtype=bf, device="cuda")
    kw = dict(a_ks=bool(aks), b_ks=bool(bks), out=out)
    extra = []
    if epi in ("bias", "gelu2", "res", "qgelu", "rope", "rope64"):
        kw["bias"] = torch.randn(N, device="cuda", generator=g).to(bf)
    if epi == "gelu2":
        pre = torch.empty((M, N), dtype=torch.uint8, device="cuda")
        kw.update(act="gelu", pre=pre, pre_deriv=True); extra.append(pre)
    if epi == "qgelu":
        kw.update(act="quick_gelu")
    if epi == "aux":
        kw.update(aux=torch.randint(0, 256, (M, N), dtype=torch.uint8, device="cuda", generator=g), dact="deriv")
    if epi == "res":
        kw.update(res=torch.randn((M, N), device="cuda", generator=g).to(bf))
    if epi == "rope":
        kw.update(rope=dict(rot=80, hd=80, period=240, span=160, L=512, log2_base=13.287712379549449))     # NeoX 4b: rotary_pct 1.0, [h][q,k,v]
    if epi == "rope64":        # a K = 1024 problem with the rotary epilogue (16 heads of 64, [h][q,k,v]): the short-K case of the persistent kernels
        kw.update(rope=dict(rot=64, hd=64, period=192, span=128, L=257, log2_base=13.287712379549449))
    return a, b, kw, out, extra


def once(a, b, kw, v, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if v in ("dwpk", "pk256", "pk128") and "b_pk" not in kw:
        return float("inf")
    e0.record()
    vv = None if v == "table" else v          # "table": the autotune table's pick for the shape (variant=None), split-K and all
    for _ in range(n):
        ops.gemm(a, b, variant=vv, **kw)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


print(f"# b = {B}; {ROUNDS} interleaved rounds; median TFLOP/s (min..max); '=' same bits as {VARIANTS[0]}, '!' differs", flush=True)
for name, M, N, K, aks, bks, epi in SHAPES:
    if FILT and FILT not in name:
        continue
    a, b, kw, out, extra = make(M, N, K, aks, bks, epi)
    if any(v in ("dwpk", "pk256", "pk128") for v in VARIANTS) and not epi.startswith("rope"):
        kw["b_pk"] = ops.pack_b(b, bool(bks))          # the pre-packed image of the same weight (frozen weights only in the product)
    fl = 2.0 * M * N * K
    vs = [v for v in VARIANTS if not (epi.startswith("rope") and v in ("w8", "pp128", "pp128x", "pp128a"))]
    ref, same, times = None, {}, {v: [] for v in vs}
    for v in vs:
        try:
            ops.gemm(a, b, variant=None if v == "table" else v, **kw); torch.cuda.synchronize()
        except Exception as e:                                   # a variant that does not serve this form
            print(f"  {name} {v}: {str(e)[:80]}"); times.pop(v); continue
        snap = (out.clone(), [x.clone() for x in extra])
        if ref is None:
            ref = snap
        same[v] = torch.equal(snap[0], ref[0]) and all(torch.equal(x, y) for x, y in zip(snap[1], ref[1]))
    n = max(2, int(2e-2 / (fl / 1.1e15)))
    for r in range(ROUNDS):
        order = list(times) if r % 2 == 0 else list(times)[::-1]
        for v in order:
            times[v].append(once(a, b, kw, v, n))
    cells = []
    for v in times:
        tf = sorted(fl / t / 1e9 for t in times[v])
        cells.append(f"{v} {statistics.median(tf):6.0f} ({tf[0]:.0f}..{tf[-1]:.0f}){'=' if same[v] else '!'}")
    print(f"{name} [{M},{N},{K}] " + " | ".join(cells), flush=True)
    del a, b, kw, out, extra
