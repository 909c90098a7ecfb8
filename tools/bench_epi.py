import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
torch.manual_seed(0)
M, N, K = 24576, 10240, 2560
a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
bias = torch.randn(N, device="cuda").to(torch.bfloat16); out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
pre = torch.empty_like(out); aux = torch.randn(M, N, device="cuda").to(torch.bfloat16)
def t(name, **kw):
    for _ in range(2): ops.gemm(a, w, out=out, variant="pp256", **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8): ops.gemm(a, w, out=out, variant="pp256", **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 8
    print(f"{name:34s} {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
t("plain")
t("bias", bias=bias)
t("bias+gelu", bias=bias, act="gelu")
t("bias+gelu+pre", bias=bias, act="gelu", pre=pre)
t("bias+gelu+pre_deriv", bias=bias, act="gelu", pre=pre, pre_deriv=True)
t("bias+pre (no act)", bias=bias, pre=pre)
t("aux dact=gelu", aux=aux, dact="gelu")
t("aux dact=deriv", aux=aux, dact="deriv")
t("res", res=aux)
