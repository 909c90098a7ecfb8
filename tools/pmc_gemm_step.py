"""PMC workload: the step's dominant GEMM kernel INSTANCES at the bench batch (64 x 512 text tokens; 512 x 257 ViT tokens), each with
its kernel variant PINNED (the committed autotune table's choice, passed explicitly: nothing is tuned inside a PMC pass), one
warm-up + three launches per case, for separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | MFMA busy + GRBM).
Writes the case manifest (label, expected kernel instance, grid size, shape, algorithmic bytes) to $PMC_MANIFEST or stdout;
tools/pmc_to_json.py joins the counter CSVs to it by (kernel instance, grid size) -- never by dispatch order."""
import json
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops            # noqa: E402

torch.manual_seed(0)
bf = torch.bfloat16
M = int(os.environ.get("PMC_M", 64 * 512))
MV = int(os.environ.get("PMC_MV", 64 * 8 * 257))
dev = "cuda"
# label, M, N, K, a_ks, b_ks, epilogue, variant, kernel-instance substring (template args as rocprofv3 prints them)
# Round 4: the kernels the autotune table picks for the b = 64 step -- the whole-row-A build (pp256a = gemm3a) in its fixed-epilogue-kind
# instantiations (last template argument: 0 PLAIN, 1 ACT, 2 AUX, 3 RES, 5 ROPE, 6 GELU2), frozen forward weights read as W^T (b_ks = 1);
# a k-strided A (weight gradients) runs the plain one-set build (gemm3x).  Two cases whose tiles would give the same (instance, grid) key
# get a row count one tile row short (M2): the counters are per launch, the row says so.
M2 = M - 256
CASES = [
    ("LM up-projection W^T + GELU + stored GELU' (KC,KS)", M, 10240, 2560, 0, 1, "out2", "pp256a", "gemm3a_bf16_kernel<false, true, 256, false, 6>"),
    ("LM dX through the down-projection x stored GELU' (KC,KS)", M, 10240, 2560, 0, 1, "aux", "pp256a", "gemm3a_bf16_kernel<false, true, 256, false, 2>"),
    ("LM down-projection W^T + bias + residual (KC,KS)", M, 2560, 10240, 0, 1, "res", "pp256a", "gemm3a_bf16_kernel<false, true, 256, false, 3>"),
    ("LM dX through the up-projection (KC,KS)", M, 2560, 10240, 0, 1, "plain", "pp256a", "gemm3a_bf16_kernel<false, true, 256, false, 0>"),
    ("LM QKV projection W^T + rotary epilogue (KC,KS)", M, 7680, 2560, 0, 1, "rope", "pp256a", "gemm3a_bf16_kernel<false, true, 256, false, 5>"),
    ("LM attention-out W^T + bias + residual (KC,KS), one tile row short", M2, 2560, 2560, 0, 1, "res", "pp256a", "gemm3a_bf16_kernel<false, true, 256, false, 3>"),
    ("gated FF up-projection + GELU + stored GELU' (KC,KC, trainable weight)", M, 10240, 2560, 0, 0, "out2", "pp256a", "gemm3a_bf16_kernel<false, false, 256, false, 6>"),
    ("gated FF dW up (KS,KS)", 10240, 2560, M, 1, 1, "plain", "pp256a", "gemm3x_bf16_kernel<true, true, 256, false"),
    ("ViT MLP up W^T + QuickGELU (KC,KS)", MV, 4096, 1024, 0, 1, "act_q", "pp256a", "gemm3a_bf16_kernel<false, true, 256, false, 1>"),
    ("ViT attention-out W^T + residual (KC,KS)", MV, 1024, 1024, 0, 1, "res", "pp256a", "gemm3a_bf16_kernel<false, true, 256, false, 3>"),
    # round 5: gemm7 (w4x; 256-thread workgroups) on the shapes the table gives it, and its L2-prefetch build next to the plain one on a long-K shape
    ("gated FF up-projection + GELU + stored GELU' (KC,KC) on w4x", M, 10240, 2560, 0, 0, "out2", "w4x", "gemm7_bf16_kernel<false, false, 6>", 256),
    ("gated FF down-projection + residual (KC,KC) on w4x", M, 2560, 10240, 0, 0, "res", "w4x", "gemm7_bf16_kernel<false, false, 3>", 256),
    ("gated FF down-projection + residual (KC,KC) on w4x + L2 prefetch", M, 2560, 10240, 0, 0, "res", "w4x_pf", "gemm7p_bf16_kernel<false, false, 3>", 256),
]
manifest = []
for case in CASES:
    label, m, n, k, aks, bks, epi, variant, inst = case[:9]
    threads = case[9] if len(case) > 9 else 512
    a = torch.randn((k, m) if aks else (m, k), device=dev).to(bf)
    b = torch.randn((k, n) if bks else (n, k), device=dev).to(bf)
    out = torch.empty((m, n), dtype=bf, device=dev)
    kw, extra = {}, 0
    if epi == "aux":
        # the product's form: the stored act'(z) as uint8 (functional.DERIV_U8)
        kw = dict(aux=torch.randint(0, 256, (m, n), device=dev, dtype=torch.uint8), dact="deriv"); extra = m * n
    elif epi == "res":
        kw = dict(bias=torch.zeros(n, device=dev, dtype=bf), res=torch.randn((m, n), device=dev).to(bf)); extra = m * n * 2
    elif epi == "act":
        kw = dict(act="gelu")
    elif epi == "act_q":
        kw = dict(bias=torch.zeros(n, device=dev, dtype=bf), act="quick_gelu")
    elif epi == "out2":
        kw = dict(bias=torch.zeros(n, device=dev, dtype=bf), act="gelu", pre=torch.empty((m, n), dtype=torch.uint8, device=dev), pre_deriv=True); extra = m * n
    elif epi == "rope":
        kw = dict(bias=torch.zeros(n, device=dev, dtype=bf), rope=dict(rot=80, hd=80, period=240, span=160, L=512, log2_base=13.287712379549449))
    for _ in range(4):                                  # 1 warm-up + 3
        ops.gemm(a, b, a_ks=bool(aks), b_ks=bool(bks), out=out, variant=variant, **kw)
    torch.cuda.synchronize()
    tiles = ((m + 255) // 256) * ((n + 255) // 256)
    wgs = min(tiles, 256) if variant in ("pp256p", "pp256px") else tiles
    manifest.append(dict(label=label, kernel=inst, grid_size=wgs * threads, shape=[m, n, k], a_kstrided=aks, b_kstrided=bks, epilogue=epi,
                         variant=variant, flop=2 * m * n * k, algorithmic_bytes=(m * k + n * k + m * n) * 2 + extra))
    del a, b, out, kw
keys = [(c["kernel"], c["grid_size"]) for c in manifest]
assert len(set(keys)) == len(keys), "two cases share (kernel instance, grid size): the counter rows could not be told apart"
path = os.environ.get("PMC_MANIFEST")
if path:
    with open(path, "w") as f:
        json.dump(manifest, f, indent=1)
else:
    print(json.dumps(manifest))
