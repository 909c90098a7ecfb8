"""ad-hoc: one-tile-per-workgroup ping-pong (pp256) vs its persistent form (pp256p) on the step's shapes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
torch.manual_seed(0)
def t(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 8 * 1e3
def run(name, m, n, k, a_ks=False, b_ks=False, **kw):
    a = torch.randn((k, m) if a_ks else (m, k), device="cuda").to(torch.bfloat16)
    b = torch.randn((k, n) if b_ks else (n, k), device="cuda").to(torch.bfloat16)
    out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    extra = {}
    if kw.get("res"): extra["res"] = torch.randn(m, n, device="cuda").to(torch.bfloat16)
    if kw.get("bias"): extra["bias"] = torch.randn(n, device="cuda").to(torch.bfloat16)
    if kw.get("act"): extra["act"] = kw["act"]
    if kw.get("aux"): extra["aux"] = torch.randn(m, n, device="cuda").to(torch.bfloat16); extra["dact"] = "deriv"
    if kw.get("pre"): extra["pre"] = torch.empty(m, n, dtype=torch.bfloat16, device="cuda"); extra["pre_deriv"] = True
    r = {v: t(lambda: ops.gemm(a, b, a_ks=a_ks, b_ks=b_ks, out=out, variant=v, **extra)) for v in ("pp256", "pp256p", "w8")}
    fl = 2.0 * m * n * k
    print(f"{name:30s} pp256 {r['pp256']:7.1f} us ({fl / r['pp256'] / 1e6:6.0f} TF)  pp256p {r['pp256p']:7.1f} us ({fl / r['pp256p'] / 1e6:6.0f} TF)  "
          f"{r['pp256'] / r['pp256p'] - 1:+.1%}   w8 {fl / r['w8'] / 1e6:6.0f} TF", flush=True)
run("ViT qkv +bias", 98688, 3072, 1024, bias=True)
run("ViT out +bias+res", 98688, 1024, 1024, bias=True, res=True)
run("ViT up +bias+qgelu", 98688, 4096, 1024, bias=True, act="quick_gelu")
run("ViT down +bias+res", 98688, 1024, 4096, bias=True, res=True)
run("LM qkv +bias", 24576, 7680, 2560, bias=True)
run("LM out +bias+res", 24576, 2560, 2560, bias=True, res=True)
run("LM up +bias+gelu+deriv", 24576, 10240, 2560, bias=True, act="gelu", pre=True)
run("LM down +bias+res", 24576, 2560, 10240, bias=True, res=True)
run("LM dgrad up (0,1) aux", 24576, 10240, 2560, b_ks=True, aux=True)
run("LM dgrad qkv (0,1)", 24576, 2560, 7680, b_ks=True)
run("wgrad (1,1) 2560x10240", 2560, 10240, 24576, a_ks=True, b_ks=True)
run("square 8192", 8192, 8192, 8192)
