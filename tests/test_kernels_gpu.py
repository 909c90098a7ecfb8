"""Per-kernel parity on a real MI355X: every C-ABI kernel against fp32 CPU math on the same seeded inputs.

Tolerances are written per test: inputs are bf16-representable, the kernels accumulate in fp32 and round the
result to bf16 once, so the bound is a few bf16 ulps (2^-8 relative) of the result scale; integer outputs are bit-exact.
"""
import math
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

bf16 = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from unimp_amd import ops as o
    return o


@pytest.fixture(params=[1, 2, 3, 4], ids=["gen1", "gen2", "gen3", "gen4"])
def attn_gen(request):
    """run an attention test on every kernel generation (1: attention.hip; 2 -- the default: attention2.hip forward + dQ, dK/dV by
    attention3.hip where it serves the form (head dim 80, causal / no mask, 32-row multiples) and by the first generation
    elsewhere; 3: attention2.hip throughout; 4: generation 2 without attention3.hip): independent implementations of one
    contract, each checked against the fp32 reference."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from unimp_amd import _lib
    old = _lib.lib().unimp_attn_set_generation(request.param)
    old3 = _lib.lib().unimp_attn_set_dkv3(2)          # generation 2 takes attention3.hip for every form it serves, also below its size threshold
    yield request.param
    _lib.lib().unimp_attn_set_dkv3(old3)
    _lib.lib().unimp_attn_set_generation(old)


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(bf16)


def close(got, want, rel=2 ** -7, name="", floor=2 ** -3):
    """PER-ROW bound (one global scale lets a kernel that is wrong on small-magnitude rows pass): the largest
    error in a row (last dimension) is bounded by rel x max(largest |want| of THAT row, floor x largest |want| of the tensor)
    -- the floor stands for the absolute error terms that do not shrink with the row (fp32 accumulation over K, bf16 rounding of
    the operands a small row shares with large ones) -- and the whole tensor's relative L2 error by rel / 2."""
    got, want = got.float().cpu(), want.float().cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    assert torch.isfinite(got).all(), name + " non-finite"
    g2 = got.reshape(-1, got.shape[-1]) if got.dim() > 1 else got.reshape(1, -1)
    w2 = want.reshape(g2.shape)
    if w2.numel() == 0:
        return
    gscale = w2.abs().max().item() + 1e-6
    rscale = w2.abs().amax(1).clamp_min(floor * gscale)
    ratio = (g2 - w2).abs().amax(1) / rscale
    r = int(ratio.argmax())
    assert ratio[r].item() <= rel, (f"{name}: row {r}: max err {(g2[r] - w2[r]).abs().max().item():.4g} vs row scale {rscale[r].item():.4g} "
                                    f"(tensor scale {gscale:.4g}; rel {ratio[r].item():.3g} > {rel:.3g})")
    l2 = ((g2 - w2).norm() / (w2.norm() + 1e-12)).item()
    assert l2 <= rel / 2, f"{name}: rel-L2 {l2:.3g} > {rel / 2:.3g}"


def act_ref(name, x):
    if name == "gelu":
        return torch.nn.functional.gelu(x)
    if name == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)
    if name == "relu":
        return torch.relu(x)
    if name == "silu":
        return torch.nn.functional.silu(x)
    return x


# ------------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 192), (300, 200, 72), (1, 8, 8), (1024, 2560, 2560),
                                   (77, 1000, 600), (513, 136, 1032),
                                   # >= 1024 rows: the 256-row LDS-DMA kernel (gemm2), incl. ragged M/N/K edges and both tile widths
                                   (1536, 1000, 600), (1304, 384, 192), (4096, 264, 128), (2048, 512, 1024), (1032, 136, 72)])
@pytest.mark.parametrize("a_ks,b_ks", [(False, False), (False, True), (True, True), (True, False)])
def test_gemm_layouts(ops, M, N, K, a_ks, b_ks):
    if (a_ks and M % 8) or (b_ks and N % 8):
        pytest.skip("k-strided operands need rows % 8 == 0")
    a, b = rnd(M, K, seed=1), rnd(N, K, seed=2)
    want = a.float() @ b.float().t()
    ad = (a.t().contiguous() if a_ks else a).cuda()
    bd = (b.t().contiguous() if b_ks else b).cuda()
    got = ops.gemm(ad, bd, a_ks=a_ks, b_ks=b_ks)                  # autotuned choice
    close(got, want, name=f"gemm {M}x{N}x{K} {a_ks}{b_ks}")
    w4x = ["w4x"] if ((not a_ks or b_ks) and K % 64 == 0 and K >= 128) else []         # gemm7.hip: whole 64-k stages; a k-strided A only with a k-strided B (dW form)
    for variant in (["v1"] if M < 256 else ["v1", "dma256", "dma128", "pp256", "pp128", "w4", "w8", "pp256p", "pp256x", "pp128x", "pp256px", "pp256a", "pp128a"] + w4x):   # every kernel, explicitly
        got = ops.gemm(ad, bd, a_ks=a_ks, b_ks=b_ks, variant=variant)
        close(got, want, name=f"gemm[{variant}] {M}x{N}x{K} {a_ks}{b_ks}")
    got32 = ops.gemm(ad, bd, a_ks=a_ks, b_ks=b_ks, out_f32=True, variant="pp256" if M >= 256 else "v1")
    close(got32, want, rel=1e-5, name="gemm f32 out")


def test_gemm_pingpong_long_k_race_screen(ops):
    """many K half-steps, odd half-step count, repeated launches: screens the ping-pong kernel's LDS ring hazards."""
    M, N = 2048, 768
    for K in (32 * 37, 32 * 128, 2560 + 8):
        a, b = rnd(M, K, seed=K), rnd(N, K, seed=K + 1)
        want = a.float() @ b.float().t()
        ad, bd = a.cuda(), b.cuda()
        ref = ops.gemm(ad, bd, variant="v1")
        close(ref, want, name="v1 long k")
        for variant in ("pp256", "pp128", "dma256", "pp256p", "pp256x", "pp128x", "pp256px", "pp256a", "pp128a"):
            outs = [ops.gemm(ad, bd, variant=variant) for _ in range(6)]
            for o in outs:
                assert torch.equal(o, outs[0]), f"{variant}: run-to-run mismatch at K={K}"
            close(outs[0], want, name=f"{variant} long k")


@pytest.mark.parametrize("pv", ["pp256p", "pp256px", "pp256x", "pp256a"])
def test_gemm_persistent_many_tiles_per_workgroup(ops, pv):
    """pp256p walks several tiles per workgroup (more tiles than CUs), ragged M / N edges, each epilogue kind; results must
    equal the one-tile-per-workgroup ping-pong kernel bit for bit (same accumulation order)."""
    M, N, K = 256 * 37 + 40, 256 * 9 + 136, 32 * 21
    a, b = rnd(M, K, seed=11), rnd(N, K, seed=12, scale=0.2)
    bias, res = rnd(N, seed=13), rnd(M, N, seed=14)
    ad, bd, biasd, resd = a.cuda(), b.cuda(), bias.cuda(), res.cuda()
    gate = torch.tensor([0.3], dtype=bf16, device="cuda")
    for kw in (dict(), dict(bias=biasd), dict(bias=biasd, act="gelu"), dict(res=resd), dict(bias=biasd, res=resd, gate=gate),
               dict(aux=resd, dact="deriv"), dict(out_f32=True)):
        want = ops.gemm(ad, bd, variant="pp256", **kw)
        for _ in range(3):
            got = ops.gemm(ad, bd, variant=pv, **kw)
            assert torch.equal(got, want), f"{pv} != pp256 for {sorted(kw)}"
    pre_w = torch.empty(M, N, dtype=bf16, device="cuda"); pre_g = torch.empty_like(pre_w)
    want = ops.gemm(ad, bd, variant="pp256", bias=biasd, act="gelu", pre=pre_w, pre_deriv=True)
    got = ops.gemm(ad, bd, variant=pv, bias=biasd, act="gelu", pre=pre_g, pre_deriv=True)
    assert torch.equal(got, want) and torch.equal(pre_g, pre_w)
    for a_ks, b_ks in ((False, True), (True, True), (True, False)):
        a2 = (a[:M - 40].t().contiguous() if a_ks else a[:M - 40]).cuda()
        b2 = (b.t().contiguous() if b_ks else b).cuda()
        want = ops.gemm(a2, b2, a_ks=a_ks, b_ks=b_ks, variant="pp256")
        got = ops.gemm(a2, b2, a_ks=a_ks, b_ks=b_ks, variant=pv)
        assert torch.equal(got, want), f"{pv} != pp256 for layout {a_ks}{b_ks}"


@pytest.mark.parametrize("M,N,K", [(256 * 5 + 40, 256 * 3 + 136, 64 * 21), (1024, 2560, 2560), (2048, 768, 64 * 37), (512, 512, 128), (300, 264, 192)])
@pytest.mark.parametrize("b_ks", [False, True])
@pytest.mark.parametrize("w4x", ["w4x", "w4x_s1", "w4x_pf", "pp256b"])
def test_gemm_w4x_equals_pingpong_bit_for_bit(ops, M, N, K, b_ks, w4x):
    """gemm7.hip (variant w4x, round 5): one wave per SIMD, 128 x 128 per wave, 64-k stages, every instruction of the main loop placed by hand
    (two fragment register sets, LDS-DMA two stages ahead, two barriers per 128 MFMAs).  Same k grouping inside every MFMA and the same k
    order per accumulator as the ping-pong kernels, same epilogue code: the results must be the ping-pong kernel's BIT FOR BIT under every
    epilogue kind, with a k-contiguous and a k-strided B, ragged M / N edges, the shortest K it serves (two stages) and long odd stage
    counts (the ring's hazards: repeated launches must agree).  Forms it does not serve are refused, not mis-served."""
    a, b = rnd(M, K, seed=21), rnd(N, K, seed=22, scale=0.2)
    bias, res = rnd(N, seed=23), rnd(M, N, seed=24)
    ad, bd, biasd, resd = a.cuda(), (b.t().contiguous() if b_ks else b).cuda(), bias.cuda(), res.cuda()
    gate = torch.tensor([0.3], dtype=bf16, device="cuda")
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    aux8 = torch.randint(0, 256, (M, N), dtype=torch.uint8, device="cuda", generator=g)
    kws = [dict(), dict(bias=biasd), dict(bias=biasd, act="gelu"), dict(bias=biasd, act="quick_gelu"), dict(res=resd), dict(bias=biasd, res=resd, gate=gate),
           dict(aux=resd, dact="deriv"), dict(out_f32=True), dict(alpha=0.125, bias=biasd), dict(gate=gate)]
    if N % 8 == 0:
        kws.append(dict(aux=aux8, dact="deriv"))
    for kw in kws:
        want = ops.gemm(ad, bd, b_ks=b_ks, variant="pp256", **kw)
        for _ in range(3):
            got = ops.gemm(ad, bd, b_ks=b_ks, variant=w4x, **kw)
            assert torch.equal(got, want), f"{w4x} != pp256 for {sorted(kw)} ({int((got != want).sum())} elements)"
    close(ops.gemm(ad, bd, b_ks=b_ks, variant=w4x), a.float() @ b.float().t(), name="w4x vs fp32")
    if N % 8 == 0:
        pre_w = torch.empty(M, N, dtype=torch.uint8, device="cuda"); pre_g = torch.empty_like(pre_w)
        want = ops.gemm(ad, bd, b_ks=b_ks, variant="pp256", bias=biasd, act="gelu", pre=pre_w, pre_deriv=True)
        got = ops.gemm(ad, bd, b_ks=b_ks, variant=w4x, bias=biasd, act="gelu", pre=pre_g, pre_deriv=True)
        assert torch.equal(got, want) and torch.equal(pre_g, pre_w)
        acc_w = ops.gemm(ad, bd, b_ks=b_ks, variant="pp256").clone(); acc_g = acc_w.clone()
        ops.gemm(ad, bd, b_ks=b_ks, variant="pp256", out=acc_w, accumulate=True)
        ops.gemm(ad, bd, b_ks=b_ks, variant=w4x, out=acc_g, accumulate=True)
        assert torch.equal(acc_g, acc_w)
    if M % 8 == 0 and not b_ks and w4x != "pp256b":
        with pytest.raises(Exception):                                   # k-strided A with a k-contiguous B is refused (UNIMP_ERR_UNSUPPORTED), not mis-served
            ops.gemm(ad.t().contiguous(), bd, a_ks=True, b_ks=False, variant=w4x)
    if M % 8 == 0 and N % 8 == 0 and b_ks and w4x == "w4x":                # the weight-gradient form: both operands k-strided
        at = ad.t().contiguous()
        for kw in (dict(), dict(gate=gate), dict(alpha=0.125), dict(out_f32=True)):
            want = ops.gemm(at, bd, a_ks=True, b_ks=True, variant="pp256", **kw)
            for _ in range(3):
                got = ops.gemm(at, bd, a_ks=True, b_ks=True, variant="w4x", **kw)
                assert torch.equal(got, want), f"w4x (dW form) != pp256 for {sorted(kw)} ({int((got != want).sum())} elements)"
        acc_w = torch.zeros(M, N, dtype=torch.float32, device="cuda"); acc_g = acc_w.clone()
        for _ in range(2):
            ops.gemm(at, bd, a_ks=True, b_ks=True, variant="pp256", out=acc_w, accumulate=True)
            ops.gemm(at, bd, a_ks=True, b_ks=True, variant="w4x", out=acc_g, accumulate=True)
        assert torch.equal(acc_g, acc_w)


@pytest.mark.parametrize("M,N,K", [(128 * 5 + 40, 256 * 3 + 136, 32 * 21), (1024, 2560, 2560), (2048, 768, 64 * 37), (512, 512, 96), (300, 264, 192), (64, 1024, 1024)])
@pytest.mark.parametrize("b_ks", [False, True])
def test_gemm_dw_two_workgroups_per_cu_equals_pingpong_bit_for_bit(ops, M, N, K, b_ks):
    """gemm9.hip (variant dw, round 6 experiment): 128 x 256 tiles, 4 waves, 72 KiB of LDS -- TWO independent workgroups per CU -- and its packed-B form
    (dwpk: the B fragments of a pre-packed frozen weight go global -> registers, three sets, no LDS).  Same k grouping inside every MFMA, same k order
    per accumulator, the ping-pong kernels' epilogue code: BIT FOR BIT the ping-pong kernel's results under every epilogue kind, both B layouts, ragged
    M / N / K (21 half-stages: the one-step tail; 3 = the shortest the packed form serves), a k-strided A (the weight-gradient form), repeated launches
    (ring hazards).  The rotary epilogue is refused."""
    a, b = rnd(M, K, seed=21), rnd(N, K, seed=22, scale=0.2)
    bias, res = rnd(N, seed=23), rnd(M, N, seed=24)
    ad, bd, biasd, resd = a.cuda(), (b.t().contiguous() if b_ks else b).cuda(), bias.cuda(), res.cuda()
    pk = ops.pack_b(bd, b_ks)
    gate = torch.tensor([0.3], dtype=bf16, device="cuda")
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    aux8 = torch.randint(0, 256, (M, N), dtype=torch.uint8, device="cuda", generator=g)
    kws = [dict(), dict(bias=biasd), dict(bias=biasd, act="gelu"), dict(bias=biasd, act="quick_gelu"), dict(res=resd), dict(bias=biasd, res=resd, gate=gate),
           dict(aux=resd, dact="deriv"), dict(out_f32=True), dict(alpha=0.125, bias=biasd), dict(gate=gate)]
    if N % 8 == 0:
        kws.append(dict(aux=aux8, dact="deriv"))
    for kw in kws:
        want = ops.gemm(ad, bd, b_ks=b_ks, variant="pp256", **kw)
        for _ in range(3):
            got = ops.gemm(ad, bd, b_ks=b_ks, variant="dw", **kw)
            assert torch.equal(got, want), f"dw != pp256 for {sorted(kw)} ({int((got != want).sum())} elements)"
            got = ops.gemm(ad, bd, b_ks=b_ks, variant="dwpk", b_pk=pk, **kw)
            assert torch.equal(got, want), f"dwpk != pp256 for {sorted(kw)} ({int((got != want).sum())} elements)"
    close(ops.gemm(ad, bd, b_ks=b_ks, variant="dw"), a.float() @ b.float().t(), name="dw vs fp32")
    if N % 8 == 0:
        pre_w = torch.empty(M, N, dtype=torch.uint8, device="cuda"); pre_g = torch.empty_like(pre_w); pre_p = torch.empty_like(pre_w)
        want = ops.gemm(ad, bd, b_ks=b_ks, variant="pp256", bias=biasd, act="gelu", pre=pre_w, pre_deriv=True)
        got = ops.gemm(ad, bd, b_ks=b_ks, variant="dw", bias=biasd, act="gelu", pre=pre_g, pre_deriv=True)
        gotp = ops.gemm(ad, bd, b_ks=b_ks, variant="dwpk", b_pk=pk, bias=biasd, act="gelu", pre=pre_p, pre_deriv=True)
        assert torch.equal(got, want) and torch.equal(pre_g, pre_w) and torch.equal(gotp, want) and torch.equal(pre_p, pre_w)
    if M % 8 == 0 and N % 8 == 0 and b_ks:                                # both operands k-strided: the weight-gradient form
        at = ad.t().contiguous()
        for kw in (dict(), dict(gate=gate), dict(out_f32=True)):
            want = ops.gemm(at, bd, a_ks=True, b_ks=True, variant="pp256", **kw)
            got = ops.gemm(at, bd, a_ks=True, b_ks=True, variant="dw", **kw)
            assert torch.equal(got, want), f"dw (dW form) != pp256 for {sorted(kw)}"
    if N % 8 == 0 and K % 8 == 0:
        with pytest.raises(Exception):
            ops.gemm(ad, bd, b_ks=b_ks, bias=biasd, rope=dict(rot=8, hd=8, period=24, span=16, L=64, log2_base=13.0), variant="dw")


def test_gemm_w4x_rotary_epilogue_equals_pingpong(ops):
    M, H, hd, L = 1024, 8, 80, 512
    a, b, bias = rnd(M, H * hd, seed=7).cuda(), rnd(3 * H * hd, H * hd, seed=8, scale=0.05).cuda(), rnd(3 * H * hd, seed=9).cuda()
    rope = dict(rot=hd, hd=hd, period=3 * hd, span=2 * hd, L=L, log2_base=float(np.log2(10000.0)))
    for bk in (False, True):
        bb = b.t().contiguous() if bk else b
        want = ops.gemm(a, bb, b_ks=bk, bias=bias, rope=rope, variant="pp256")
        got = ops.gemm(a, bb, b_ks=bk, bias=bias, rope=rope, variant="w4x")
        assert torch.equal(got, want), int((got != want).sum())


def test_gemm_variants_same_bits_under_every_forward_epilogue(ops):
    """The kernel variant is chosen per (M, N, K) -- from the autotune table, or by timing on a shape the table lacks -- so a sample
    run alone (M = L) and inside a batch (M = B L) may go through different kernels: every variant must give the SAME BITS under every
    epilogue the forward pass uses.  Round 3: `v * tanh(gate) + res` compiled to v_mul + v_add in the 128 x 128 / LDS-DMA kernels and
    to one v_fmac in the ping-pong kernels (HIP's __fmul_rn is a plain, contractable `x * y`): ~4 elements per million one bf16 ulp
    apart, a once-per-ten-boxes failure of the per-sample-independence tests.  The gated cross-attention's shapes, and the rotary
    epilogue's two kernels."""
    gate = torch.tensor([0.5], dtype=bf16, device="cuda")
    for M, N, K in ((512, 2560, 512), (1024, 2560, 2048), (640, 1032, 416)):
        a, b = rnd(M, K, seed=M + K).cuda(), rnd(N, K, seed=N + K, scale=0.05).cuda()
        bias, res = rnd(N, seed=3).cuda(), rnd(M, N, seed=4).cuda()
        for kw in (dict(), dict(bias=bias), dict(res=res), dict(res=res, gate=gate), dict(bias=bias, res=res, gate=gate), dict(gate=gate), dict(act="gelu"),
                   dict(bias=bias, act="gelu"), dict(bias=bias, act="quick_gelu"), dict(alpha=0.125, bias=bias), dict(alpha=0.125, res=res, gate=gate)):
            outs = {v: ops.gemm(a, b, variant=v, **kw) for v in ("v1", "dma256", "dma128", "pp256", "pp128", "w8", "pp256p", "pp256x", "pp128x", "pp256px", "pp256a", "pp128a")}
            for v, o in outs.items():
                assert torch.equal(o, outs["v1"]), f"[{M}, {N}, {K}] {sorted(kw)}: {v} differs from v1 in {int((o != outs['v1']).sum())} elements"
    M, H, hd, L = 1024, 8, 80, 512
    a, b, bias = rnd(M, H * hd, seed=7).cuda(), rnd(3 * H * hd, H * hd, seed=8, scale=0.05).cuda(), rnd(3 * H * hd, seed=9).cuda()
    rope = dict(rot=hd, hd=hd, period=3 * hd, span=2 * hd, L=L, log2_base=float(np.log2(10000.0)))
    o4, o9 = ops.gemm(a, b, bias=bias, rope=rope, variant=4), ops.gemm(a, b, bias=bias, rope=rope, variant=9)
    assert torch.equal(o4, o9), f"rotary epilogue: pp256 and pp256p differ in {int((o4 != o9).sum())} elements"


def test_forward_gemm_rows_do_not_depend_on_the_row_count(ops):
    """A sample's rows of a FORWARD projection must come out with the same bits whatever the batch around them: no split-K outside a
    backward pass (its slice count follows the tile count), and every variant the tuner may pick per M sums k in the same order.
    The gated cross-attention's to_q ([B L, 512] x K = 2560: 16 tiles at b = 1, 96 at b = 6, 384 at b = 24 -- round 3 found it under
    5 / 3 / 0 slices), a Perceiver-sized and an LM-sized projection."""
    for N, K in ((512, 2560), (1024, 4096), (2560, 2560)):
        a, b = rnd(12288, K, seed=N).cuda(), rnd(N, K, seed=K, scale=0.05).cuda()
        want = ops.gemm(a[:512], b)
        for M in (1024, 3072, 6144, 12288):
            got = ops.gemm(a[:M], b)
            assert torch.equal(got[:512], want), f"[{M}, {N}, {K}]: the first 512 rows differ from the M = 512 call in {int((got[:512] != want).sum())} elements"


def test_forward_gemm_recomputed_inside_a_backward_pass_keeps_its_bits(ops):
    """split-K eligibility is declared by the backward bodies (ops.backward_scope), not sniffed from the autograd engine: a
    FORWARD GEMM that runs while autograd's backward is executing (activation checkpointing recomputes the forward there) takes the
    forward's path and produces the forward's bits; the same call from a declared backward body may split."""
    a, b = rnd(1024, 2560, seed=7).cuda(), rnd(512, 2560, seed=8, scale=0.05).cuda()      # the gated cross-attention's to_q: 16 tiles
    want = ops.gemm(a, b)
    seen = {}

    class Recompute(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 1.0

        @staticmethod
        def backward(ctx, dy):                      # NOT a backward_scope body: this is a recomputed forward
            seen["recomputed"] = ops.gemm(a, b)
            return dy

    x = torch.ones(4, device="cuda", requires_grad=True)
    Recompute.apply(x).sum().backward()
    assert torch.equal(seen["recomputed"], want)
    assert ops._IN_BACKWARD == 0
    inside = ops.backward_scope(lambda: ops._IN_BACKWARD)()
    assert inside == 1 and ops._IN_BACKWARD == 0


@pytest.mark.parametrize("M,N,K,act", [(512, 1024, 256, "gelu"), (300, 520, 192, "gelu"), (1024, 2560, 512, "quick_gelu"), (128, 264, 128, "gelu")])
def test_gemm_stored_derivative_uint8(ops, M, N, K, act):
    """act'(z) in 8 bits (pre_deriv = 2 / dact = 6: q = round(202 g + 27)): the up-projection's second output and the backward's aux
    operand at half the bytes.  Every variant writes the same bytes; they decode to the bf16 form's derivative within half a step
    (0.0025) plus bf16's own rounding; 0 and 1 are exact (a ReLU'd copy); the backward product through the uint8 operand equals the
    one through the decoded values, and is as close to fp32 as the bf16 form's."""
    a, b, bias = rnd(M, K, seed=1).cuda(), rnd(N, K, seed=2, scale=0.2).cuda(), rnd(N, seed=3).cuda()
    ld = (N + 7) // 8 * 8
    outs = {}
    for v in (["v1"] if M < 256 else ["v1", "pp256", "pp128", "w8", "pp256p", "pp256x", "pp128x", "pp256px", "pp256a", "pp128a"]):       # the kernels with the specialised epilogue kinds
        q = torch.full((M, ld), 255, dtype=torch.uint8, device="cuda")
        y = ops.gemm(a, b, bias=bias, act=act, pre=q[:, :N], pre_deriv=True, variant=v)
        outs[v] = (y, q)
    y0, q0 = outs["v1"]
    for v, (y, q) in outs.items():
        assert torch.equal(y, y0) and torch.equal(q, q0), f"{v}: {int((q != q0).sum())} bytes differ from v1"
    assert torch.all(q0[:, N:] == 255), "wrote past column N"
    g16 = torch.empty((M, ld), dtype=bf16, device="cuda")
    ops.gemm(a, b, bias=bias, act=act, pre=g16[:, :N], pre_deriv=True, variant="v1")
    dec = (q0[:, :N].float() - 27.0) / 202.0
    err = (dec - g16[:, :N].float()).abs()
    assert float(err.max()) <= 0.5 / 202 + 2 ** -8 + 1e-6, float(err.max())
    z = (a.float() @ b.float().t() + bias.float())
    dead, sat = z < -8.0, z > 8.0
    assert torch.all(dec[dead] == 0.0) and torch.all(dec[sat] == 1.0)
    # backward: dz = (dy W2) * g through the uint8 operand
    dy, w2 = rnd(M, 96, seed=5).cuda(), rnd(96, N, seed=6, scale=0.2).cuda()
    want = (dy.float() @ w2.float()) * dec
    for v in (["v1"] if M < 256 else ["v1", "pp256", "pp128", "w8", "pp256p", "pp256x", "pp128x", "pp256px", "pp256a", "pp128a"]):
        got = ops.gemm(dy, w2, b_ks=True, aux=q0[:, :N], dact="deriv", variant=v)
        close(got, want, name=f"dz through uint8 act' [{v}]")
    ref16 = ops.gemm(dy, w2, b_ks=True, aux=g16[:, :N], dact="deriv", variant="v1")
    zc = z.cpu().double()
    if act == "gelu":
        gt = 0.5 * (1 + torch.erf(zc / 2 ** 0.5)) + zc * torch.exp(-zc * zc / 2) / (2 * math.pi) ** 0.5
    else:
        sg = torch.sigmoid(1.702 * zc)
        gt = sg * (1 + 1.702 * zc * (1 - sg))
    truth = ((dy.float() @ w2.float()).cpu().double() * gt).float()
    got = ops.gemm(dy, w2, b_ks=True, aux=q0[:, :N], dact="deriv", variant="v1").float().cpu()
    from unimp_amd._lib import UnimpHipError
    with pytest.raises(UnimpHipError):           # a kernel without the uint8 forms refuses instead of reading the bytes as bf16
        ops.gemm(dy, w2, b_ks=True, aux=q0[:, :N], dact="deriv", variant="dma256")
    e8 = float((got - truth).norm() / truth.norm())
    e16 = float((ref16.float().cpu() - truth).norm() / truth.norm())
    assert e8 <= max(2.0 * e16, 6e-3), (e8, e16)


def test_stored_derivative_uint8_decodes_0_and_1_exactly(ops):
    """KAT for common.h deriv_u8_get: byte 27 is EXACTLY 0 (a dead ReLU / GELU unit passes no gradient, not -7e-9 dy) and
    byte 229 is exactly 1 (the product equals the plain GEMM bit for bit), in every kernel that serves the uint8 operand."""
    M, N, K = 512, 512, 256
    dy, w2 = rnd(M, K, seed=41).cuda(), rnd(K, N, seed=42, scale=0.2).cuda()
    for v in ("v1", "pp256", "pp128", "w8", "pp256p", "pp256x", "pp128x", "pp256px", "pp256a", "pp128a"):
        plain = ops.gemm(dy, w2, b_ks=True, variant=v)
        zero = ops.gemm(dy, w2, b_ks=True, aux=torch.full((M, N), 27, dtype=torch.uint8, device="cuda"), dact="deriv", variant=v)
        one = ops.gemm(dy, w2, b_ks=True, aux=torch.full((M, N), 229, dtype=torch.uint8, device="cuda"), dact="deriv", variant=v)
        assert torch.all(zero == 0), f"{v}: byte 27 leaks {float(zero.float().abs().max())}"
        assert torch.equal(one, plain), f"{v}: byte 229 is not exactly 1"


def test_gemm_ragged_n_padded_rows(ops):
    """N % 8 != 0 with row strides padded to a multiple of 8 (the 74 053-column LM head): full 8-column groups take the vector
    epilogue, the last partial group is written element by element; nothing beyond column N - 1 is touched."""
    M, N, K, ld = 1100, 1003, 200, 1008
    a, b, bias = rnd(M, K, seed=21), rnd(N, K, seed=22, scale=0.2), rnd(N, seed=23)
    res = rnd(M, N, seed=24)
    z = a.float() @ b.float().t() + bias.float()
    resp = torch.zeros(M, ld, dtype=bf16); resp[:, :N] = res
    for variant in ("pp256", "pp128", "w8", "pp256p", "v1"):
        for kw, want in ((dict(), z), (dict(act="gelu"), act_ref("gelu", z)), (dict(res=resp.cuda()[:, :N]), z + res.float())):
            buf = torch.full((M, ld), 7.0, dtype=bf16, device="cuda")
            got = ops.gemm(a.cuda(), b.cuda(), bias=bias.cuda(), out=buf[:, :N], variant=variant, **kw)
            close(got, want, name=f"ragged N [{variant}] {sorted(kw)}")
            assert torch.all(buf[:, N:] == 7.0), f"{variant}: wrote past column N"
        pre = torch.full((M, ld), 7.0, dtype=bf16, device="cuda")
        buf = torch.full((M, ld), 7.0, dtype=bf16, device="cuda")
        ops.gemm(a.cuda(), b.cuda(), bias=bias.cuda(), act="gelu", pre=pre[:, :N], out=buf[:, :N], variant=variant)
        close(pre[:, :N], z, name=f"ragged N pre [{variant}]")
        assert torch.all(pre[:, N:] == 7.0) and torch.all(buf[:, N:] == 7.0)
    f32 = torch.full((M, ld), 7.0, dtype=torch.float32, device="cuda")
    got = ops.gemm(a.cuda(), b.cuda(), out=f32[:, :N], variant="pp256")
    close(got, a.float() @ b.float().t(), rel=1e-5, name="ragged N f32")
    assert torch.all(f32[:, N:] == 7.0)


def test_gemm_asymmetric_identity(ops):
    """A = I with an asymmetric B catches a transposed C write (cdna guide §3)."""
    n = 128
    a = torch.eye(n).to(bf16)
    b = (torch.arange(n)[:, None] * 2 + torch.arange(n)[None, :] % 7).float().to(bf16)   # asymmetric
    got = ops.gemm(a.cuda(), b.cuda())          # C = A @ B^T = B^T
    assert torch.equal(got.float().cpu(), b.float().t())


@pytest.mark.parametrize("M", [200, 1100])
@pytest.mark.parametrize("act", [None, "gelu", "quick_gelu", "relu"])
def test_gemm_epilogue_bias_act_pre(ops, act, M):
    N, K = 264, 136
    a, b, bias = rnd(M, K, seed=3), rnd(N, K, seed=4, scale=0.2), rnd(N, seed=5)
    z = a.float() @ b.float().t() + bias.float()
    for variant in (["v1"] if M < 256 else ["v1", "dma128", "pp256", "pp128", "pp256p", "pp256x", "pp128x", "pp256px", "pp256a", "pp128a"]):
        pre = torch.empty(M, N, dtype=bf16, device="cuda")
        got = ops.gemm(a.cuda(), b.cuda(), bias=bias.cuda(), act=act, pre=pre, variant=variant)
        close(pre, z, name="pre")
        close(got, act_ref(act, z), name=f"act {act} [{variant}]")
        if act is not None:                         # forward epilogue that stores act'(z) instead of z, consumed by dact="deriv"
            dz = torch.empty(M, N, dtype=bf16, device="cuda")
            got = ops.gemm(a.cuda(), b.cuda(), bias=bias.cuda(), act=act, pre=dz, pre_deriv=True, variant=variant)
            close(got, act_ref(act, z), name=f"act {act} [{variant}] with deriv")
            zz = z.clone().requires_grad_(True)
            act_ref(act, zz).sum().backward()
            close(dz, zz.grad, name=f"stored act'(z) {act} [{variant}]")
            up = rnd(M, N, seed=9)
            w2 = rnd(N, N, seed=10, scale=0.1)
            got = ops.gemm(up.cuda(), w2.cuda(), b_ks=True, aux=dz, dact="deriv", variant=variant)
            close(got, (up.float() @ w2.float()) * dz.float().cpu(), name="dact=deriv")


@pytest.mark.parametrize("M", [136, 1160])
def test_gemm_epilogue_gate_res_dact_accum(ops, M):
    N, K = 200, 264
    a, b = rnd(M, K, seed=6), rnd(N, K, seed=7, scale=0.2)
    res, aux = rnd(M, N, seed=8), rnd(M, N, seed=9)
    gate = torch.tensor([0.7]).to(bf16)
    z = a.float() @ b.float().t()
    want = z * math.tanh(float(gate.float())) + res.float()
    got = ops.gemm(a.cuda(), b.cuda(), gate=gate.cuda(), res=res.cuda())
    close(got, want, name="gate+res")
    bias = rnd(N, seed=10)
    for variant in (["v1"] if M < 256 else ["v1", "pp256", "pp128", "w8", "pp256p", "dma128"]):   # gated block: raw (pre-gate) second output
        raw = torch.empty(M, N, dtype=bf16, device="cuda")
        got = ops.gemm(a.cuda(), b.cuda(), bias=bias.cuda(), gate=gate.cuda(), res=res.cuda(), pre=raw, variant=variant)
        close(raw, z + bias.float(), name=f"raw [{variant}]")
        close(got, (z + bias.float()) * math.tanh(float(gate.float())) + res.float(), name=f"gate+res+raw [{variant}]")
    tg = math.tanh(float(gate.float()))
    for variant in (["v1"] if M < 256 else ["v1", "pp256", "pp128", "w8", "pp256p", "dma128"]):   # tanh(gate) without a residual: the
        got = ops.gemm(a.cuda(), b.cuda(), gate=gate.cuda(), variant=variant)                      # gated block's dX / dW GEMMs
        close(got, z * tg, name=f"gate only [{variant}]")
        got = ops.gemm(a.cuda(), b.cuda(), gate=gate.cuda(), aux=aux.cuda(), dact="deriv", variant=variant)
        close(got, z * aux.float() * tg, name=f"aux deriv + gate [{variant}]")
        got = ops.gemm(a.cuda(), b.cuda(), bias=bias.cuda(), act="gelu", gate=gate.cuda(), variant=variant)
        close(got, act_ref("gelu", z + bias.float()) * tg, name=f"act + gate [{variant}]")
    x = aux.float().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    got = ops.gemm(a.cuda(), b.cuda(), aux=aux.cuda(), dact="gelu")
    close(got, z * x.grad, name="dact gelu")
    x = aux.float().requires_grad_(True)
    (x * torch.sigmoid(1.702 * x)).sum().backward()
    got = ops.gemm(a.cuda(), b.cuda(), aux=aux.cuda(), dact="quick_gelu")
    close(got, z * x.grad, name="dact quick_gelu")
    acc = torch.full((M, N), 3.0, dtype=torch.float32, device="cuda")
    ops.gemm(a.cuda(), b.cuda(), out=acc, accumulate=True, alpha=0.5)
    close(acc, 0.5 * z + 3.0, rel=1e-5, name="accumulate f32")


@pytest.mark.parametrize("M", [1, 10, 16, 17, 40, 64])
@pytest.mark.parametrize("N,K", [(256, 64), (1005, 192), (2560, 2560), (24, 1344),
                                 (7680, 1024), (5133, 1088), (17000, 1024)])      # two / four weight tiles per block (ragged last block), 16 waves at M <= 16
def test_gemm_skinny_decode_rows(ops, M, N, K):
    """the weight-streaming kernel used for decode rows (M <= 64): against fp32, against the 128x128 kernel, every
    epilogue flavour the decode GEMMs use (bias, act, gate, residual), ragged N with a padded ldc."""
    a, b = rnd(M, K, seed=21), rnd(N, K, seed=22, scale=0.2)
    bias, res = rnd(N, seed=23), rnd(M, N, seed=24)
    gate = torch.tensor([0.4]).to(bf16)
    z = a.float() @ b.float().t()
    ad, bd = a.cuda(), b.cuda()
    ldc = (N + 7) // 8 * 8
    got = ops.gemm(ad, bd, ldc=ldc)                                  # routed to the skinny kernel by shape
    close(got, z, name="skinny plain")
    ref = ops.gemm(ad, bd, ldc=ldc, variant="v1")
    close(got, ref.float(), rel=1e-2, name="skinny vs v1")
    got = ops.gemm(ad, bd, variant="skinny", bias=bias.cuda(), act="gelu")
    close(got, torch.nn.functional.gelu(z + bias.float()), name="skinny bias+gelu")
    got = ops.gemm(ad, bd, variant="skinny", bias=bias.cuda(), res=res.cuda())
    close(got, z + bias.float() + res.float(), name="skinny bias+res")
    got = ops.gemm(ad, bd, variant="skinny", gate=gate.cuda(), res=res.cuda())
    close(got, z * math.tanh(float(gate.float())) + res.float(), name="skinny gate+res")
    got = ops.gemm(ad, bd, variant="skinny", out_f32=True)
    close(got, z, rel=1e-5, name="skinny f32")


@pytest.mark.parametrize("M", [1, 5, 8, 10, 16])
@pytest.mark.parametrize("N,K", [(2560, 2560), (7680, 2560), (10240, 2560), (2560, 10240), (2560, 512), (512, 2560), (74053, 2560), (4096, 4096), (2052, 6400), (4099, 8192),
                                 (1005, 1024), (24, 512), (5133, 1536), (777, 3072), (640, 64), (512, 2624), (1024, 16384), (520, 4160), (40, 7744)])
def test_gemm_skinny2_persistent_decode_rows(ops, M, N, K):
    """round 6: the second-generation weight-streaming kernel (gemm.hip skinny2: every load of a wave issued before its first wait, 8 waves x
    up to five 64-k chunks or 16 x up to four) that serves M <= 16 decode rows with K <= 4096 -- the cfg2 / cfg5 decode shapes, the head's
    4 629 tiles, N = 24, ragged N with a padded ldc, every chunk count per wave incl. a ragged last one (K = 1536: 24 chunks on 8 waves) -- and its
    long-K form (K > 4096: rounds of three chunks per wave, the next round's loads issued before this round's MFMAs; K = 10 240, 16 384, a
    ragged 4 160 and 7 744); the weight rows per workgroup follow N (the busiest CU's rows are minimised: 10 for N = 2560 and 10 240, 15 for
    7680, 16 for the head, 6 for N = 24, ragged last workgroups at N = 1005 / 777 / 2052 / 4099) -- against fp32 and against the round-3 kernel on the same operands (different partial-sum
    grouping: bf16-level agreement), all epilogue flavours of the decode step; launching twice gives the same bits (fixed summation order)."""
    from unimp_amd import _lib
    L = _lib.lib()
    a, b = rnd(M, K, seed=31), rnd(N, K, seed=32, scale=0.2)
    bias, res = rnd(N, seed=33), rnd(M, N, seed=34)
    gate = torch.tensor([0.4]).to(bf16)
    z = a.float() @ b.float().t()
    ad, bd = a.cuda(), b.cuda()
    ldc = (N + 7) // 8 * 8
    assert L.unimp_gemm_set_skinny2(1) in (0, 1)
    try:
        got = ops.gemm(ad, bd, ldc=ldc)
        close(got, z, name="skinny2 plain")
        again = ops.gemm(ad, bd, ldc=ldc)
        assert torch.equal(got, again), "skinny2: two launches differ"
        L.unimp_gemm_set_skinny2(0)
        old = ops.gemm(ad, bd, ldc=ldc)
        L.unimp_gemm_set_skinny2(1)
        close(got, old.float(), rel=1e-2, name="skinny2 vs the round-3 kernel")
        got = ops.gemm(ad, bd, variant="skinny", bias=bias.cuda(), act="gelu")
        close(got, torch.nn.functional.gelu(z + bias.float()), name="skinny2 bias+gelu")
        got = ops.gemm(ad, bd, variant="skinny", bias=bias.cuda(), res=res.cuda())
        close(got, z + bias.float() + res.float(), name="skinny2 bias+res")
        got = ops.gemm(ad, bd, variant="skinny", gate=gate.cuda(), res=res.cuda())
        close(got, z * math.tanh(float(gate.float())) + res.float(), name="skinny2 gate+res")
        got = ops.gemm(ad, bd, variant="skinny", out_f32=True)
        close(got, z, rel=1e-5, name="skinny2 f32")
    finally:
        L.unimp_gemm_set_skinny2(1)


@pytest.mark.parametrize("M", [1, 2, 3, 6, 10, 13, 16])
@pytest.mark.parametrize("N,K,beta", [(7680, 2560, True), (10240, 2560, True), (512, 2560, True), (74053, 2560, True), (12288, 4096, False),
                                      (4096, 1024, True), (1005, 512, False), (333, 1536, True), (256, 3072, False), (128, 2048, True)])
def test_gemm_skinny2_fused_layernorm(ops, M, N, K, beta):
    """ops.gemm(ln=...): the LayerNorm in front of a decode-step projection runs inside the weight-streaming GEMM (gemm_skinny2_ln_kernel: the
    workgroup normalises the M real rows once, in 512-element units spread over its waves, and the waves read their MFMA fragments from LDS;
    every K it takes -- 512 ... 3072 and 4096 -- and every units-per-wave form: M = 1 ... 16).  Against the two-launch
    form on the same operands (unimp_layernorm_fwd then the same kernel: the normalised rows differ at most by the statistics' summation
    order, i.e. a bf16 ulp on a few elements) and against fp32 math; with and without beta (MPT's LayerNorm has none); bias + GELU on top
    (the up-projection); launching twice gives the same bits; refusals: K beyond 4096 or not in whole 512s, more than 16 rows, and the one
    shape whose row image does not fit the CU's LDS (K = 4096 with 15 or 16 rows)."""
    x, w = rnd(M, K, seed=41, scale=2.0) + 0.5, rnd(N, K, seed=42, scale=0.05)
    g, b_, bias = (rnd(K, seed=43, scale=0.2) + 1.0).to(bf16), rnd(K, seed=44, scale=0.3), rnd(N, seed=45)
    eps = 1e-5
    xd, wd, gd, bd = x.cuda(), w.cuda(), g.cuda(), (b_.cuda() if beta else None)
    if K == 4096 and M == 16:
        assert not ops.skinny_ln_ok(M, K)
        with pytest.raises(Exception):
            ops.gemm(xd, wd, ln=(gd, bd, eps))
        return
    assert ops.skinny_ln_ok(M, K)
    h, _, _ = ops.layernorm_fwd(xd, gd, bd, eps)
    want2 = ops.gemm(h, wd)
    got = ops.gemm(xd, wd, ln=(gd, bd, eps))
    hf = torch.nn.functional.layer_norm(x.float(), (K,), g.float(), b_.float() if beta else None, eps).to(bf16).float()
    close(got, hf @ w.float().t(), name="fused LN vs fp32")
    close(got, want2.float(), rel=2 ** -8, name="fused LN vs layernorm_fwd + gemm")
    assert torch.equal(got, ops.gemm(xd, wd, ln=(gd, bd, eps))), "fused LN: two launches differ"
    got = ops.gemm(xd, wd, bias=bias.cuda(), act="gelu", ln=(gd, bd, eps))
    close(got, torch.nn.functional.gelu(hf @ w.float().t() + bias.float()), name="fused LN + bias + gelu")
    assert not ops.skinny_ln_ok(17, K) and not ops.skinny_ln_ok(M, 8192) and not ops.skinny_ln_ok(M, 2560 + 64) and not ops.skinny_ln_ok(M, 3584)
    with pytest.raises(Exception):
        ops.gemm(rnd(32, K, seed=1).cuda(), wd, ln=(gd, bd, eps))


@pytest.mark.parametrize("R,nh,hd,rot,interleaved", [(1, 32, 80, 80, True), (10, 32, 80, 80, True), (5, 8, 64, 32, True), (3, 32, 128, 0, False),
                                                     (16, 12, 64, 64, False)])
def test_decode_rope_append(ops, R, nh, hd, rot, interleaved):
    """unimp_decode_rope_append: one launch = rope_ on the step's q / k rows (row r at its own position) + the two index_put_ of the rotated k
    and of v into slot pos_idx[r] of the caches -- bit-identical with that three-launch sequence (GPT-NeoX interleaved [nh, 3 hd] and the
    [3, nh, hd] layout; partial rotation; rot = 0 = append only: MPT / OPT); nothing else in the caches moves."""
    H, cap = nh * hd, 40
    g = torch.Generator().manual_seed(R * 7 + hd)
    qkv = torch.randn(R, 3 * H, generator=g).to(bf16).cuda()
    pos = torch.randint(0, cap, (R,), generator=g).cuda()
    kc = torch.randn(R, cap, nh, hd, generator=g).to(bf16).cuda()
    vc = torch.randn(R, cap, nh, hd, generator=g).to(bf16).cuda()
    half = rot // 2
    cos = torch.rand(R, max(half, 1), generator=g).cuda() if rot else None
    sin = torch.rand(R, max(half, 1), generator=g).cuda() if rot else None
    if interleaved:
        hs, offs = 3 * hd, (0, hd, 2 * hd)
        view = lambda t: t.view(R, 1, nh, 3 * hd)
        kv = lambda t: (view(t)[..., hd:2 * hd], view(t)[..., 2 * hd:])
    else:
        hs, offs = hd, (0, H, 2 * H)
        kv = lambda t: (t.view(R, 1, 3, nh, hd)[:, :, 1], t.view(R, 1, 3, nh, hd)[:, :, 2])
    want_qkv, want_k, want_v = qkv.clone(), kc.clone(), vc.clone()
    if rot:
        ops.rope_(want_qkv, R, nh, hs, rot, offs[:2], cos, sin)
    k_, v_ = kv(want_qkv)
    rows = torch.arange(R, device="cuda")
    want_k.index_put_((rows, pos), k_[:, 0])
    want_v.index_put_((rows, pos), v_[:, 0])
    assert ops.decode_rope_append_ok(hd, rot, hs, offs, qkv, kc)
    ops.decode_rope_append(qkv, nh, hs, hd, offs, rot, cos, sin, kc, vc, pos)
    torch.cuda.synchronize()
    assert torch.equal(qkv, want_qkv), "q / k rows"
    assert torch.equal(kc, want_k) and torch.equal(vc, want_v), "cache slots"


@pytest.mark.parametrize("R,nh,hd,rot,interleaved,cap,alibi", [(1, 32, 80, 80, True, 530, False), (10, 32, 80, 80, True, 530, False), (5, 8, 64, 32, True, 300, False),
                                                               (3, 32, 128, 0, False, 1100, True), (16, 12, 64, 64, False, 90, False), (2, 4, 80, 80, True, 2100, True)])
def test_attn_decode_step_one_launch(ops, R, nh, hd, rot, interleaved, cap, alibi):
    """unimp_attn_decode_step: the self-attention of a cached decode step in one launch -- against unimp_decode_rope_append followed by the
    split-key unimp_attn_decode on the same operands: the cache afterwards holds the same bits (rotated k, v in slot pos[r]; nothing else moves),
    the outputs are bit-identical (same key chunks, same merge order; the last workgroup of a (row, head) merges, handed the partials through
    agent-scope accesses) and agree with fp32 math; rows sit at different positions incl. 0 (one key: the new one) and the last slot; GPT-NeoX
    interleaved and [3, nh, hd] layouts, partial rotation, no rotation + ALiBi (MPT), head dims 64 / 80 / 128, 1 ... 17 key chunks; fifty more
    launches give the same bits (the hand-over has no lucky timing); a row alone gives the bits it gives in the batch."""
    H = nh * hd
    g = torch.Generator().manual_seed(R * 11 + hd + cap)
    qkv = torch.randn(R, 3 * H, generator=g).to(bf16).cuda()
    pos = torch.randint(0, cap, (R,), generator=g)
    pos[0] = cap - 1
    if R > 1:
        pos[1] = 0
    pos = pos.cuda()
    kc = torch.randn(R, cap, nh, hd, generator=g).to(bf16).cuda()
    vc = torch.randn(R, cap, nh, hd, generator=g).to(bf16).cuda()
    half = rot // 2
    cos = torch.rand(R, max(half, 1), generator=g).cuda() if rot else None
    sin = torch.rand(R, max(half, 1), generator=g).cuda() if rot else None
    slopes = (0.5 ** torch.arange(1, nh + 1).float() * 4).cuda() if alibi else None
    hs, offs = (3 * hd, (0, hd, 2 * hd)) if interleaved else (hd, (0, H, 2 * H))
    scale = hd ** -0.5
    # the three-launch form
    q2, k2, v2 = qkv.clone(), kc.clone(), vc.clone()
    ops.decode_rope_append(q2, nh, hs, hd, offs, rot, cos, sin, k2, v2, pos)
    qv = q2.view(R, 1, nh, 3 * hd)[..., :hd] if interleaved else q2.view(R, 1, 3, nh, hd)[:, :, 0]
    want = ops.attn_decode(qv, k2, v2, scale, (pos + 1).int(), slopes)
    # one launch
    q1, k1, v1 = qkv.clone(), kc.clone(), vc.clone()
    got = ops.attn_decode_step(q1, nh, hs, hd, offs, rot, cos, sin, k1, v1, pos, scale, slopes)
    torch.cuda.synchronize()
    assert torch.equal(q1, qkv), "qkv must stay as it was"
    assert torch.equal(k1, k2) and torch.equal(v1, v2), "cache after the step"
    assert torch.equal(got, want), "one launch vs rope_append + split-key decode: same chunks, same orders, same bits"
    # fp32 math on the cache the step left behind
    kf, vf, qf = k2.float(), v2.float(), qv.float()[:, 0]
    for r in range(R):
        n = int(pos[r]) + 1
        sc = torch.einsum("hd,khd->hk", qf[r], kf[r, :n]) * scale
        if alibi:
            sc = sc + slopes[:, None] * torch.arange(n, device="cuda")[None, :].float()
        ref = torch.einsum("hk,khd->hd", torch.softmax(sc, -1), vf[r, :n])
        close(got[r, 0], ref, name=f"row {r} vs fp32")
    k3, v3 = kc.clone(), vc.clone()
    for _ in range(50):
        again = ops.attn_decode_step(qkv, nh, hs, hd, offs, rot, cos, sin, k3, v3, pos, scale, slopes)
        assert torch.equal(got, again), "two launches differ"
    r = R - 1
    alone = ops.attn_decode_step(qkv[r:r + 1].clone(), nh, hs, hd, offs, rot, cos[r:r + 1].contiguous() if rot else None, sin[r:r + 1].contiguous() if rot else None,
                                 kc[r:r + 1].clone(), vc[r:r + 1].clone(), pos[r:r + 1].contiguous(), scale, slopes)
    assert torch.equal(alone[0], got[r]), "a row alone and in the batch"


@pytest.mark.parametrize("K,groups,nh,hd,rot,cap", [(10, 1, 32, 80, 80, 530), (4, 3, 8, 64, 32, 300), (16, 1, 4, 128, 0, 700), (7, 2, 12, 64, 64, 140), (2, 2, 4, 80, 80, 1100)])
def test_attn_decode_step_grouped_beams(ops, K, groups, nh, hd, rot, cap):
    """unimp_attn_decode_step with beam groups: the prompt's keys (identical in the K rows of a group below shared_len[g]) are read once per prompt, in
    chunks of 32 keys by the prefix workgroups of the same launch; the rows' own tails and the new key by the tail workgroups; a row's arrival target
    is the number of workgroups that hold keys of it (derived from shared_len / pos on the device: empty workgroups neither publish nor arrive).
    Against the ungrouped one-launch form on the same operands (another partition of the keys: fp32 rounding) and fp32 math; the same cache
    afterwards, bit for bit; groups sit at different prompt lengths incl. one that ends on a chunk boundary; tails that cross a 128-key boundary;
    1 ... 4 queries per prefix wave; thirty more launches give the same bits."""
    R, H = K * groups, nh * hd
    g = torch.Generator().manual_seed(K * 13 + hd + cap)
    qkv = torch.randn(R, 3 * H, generator=g).to(bf16).cuda()
    shared = torch.tensor([(cap - 60 - 17 * i) // (32 if i == 1 else 1) * (32 if i == 1 else 1) for i in range(groups)], dtype=torch.int32)
    done = torch.randint(0, 50, (groups,), generator=g)
    pos = (shared.long() + done).repeat_interleave(K).cuda()
    kc = torch.randn(R, cap, nh, hd, generator=g).to(bf16)
    vc = torch.randn(R, cap, nh, hd, generator=g).to(bf16)
    for gi in range(groups):                                  # the beams of a prompt share the prompt's K / V
        n = int(shared[gi])
        kc[gi * K:(gi + 1) * K, :n] = kc[gi * K, :n]
        vc[gi * K:(gi + 1) * K, :n] = vc[gi * K, :n]
    kc, vc, shared = kc.cuda(), vc.cuda(), shared.cuda()
    half = rot // 2
    cos = torch.rand(R, max(half, 1), generator=g).cuda() if rot else None
    sin = torch.rand(R, max(half, 1), generator=g).cuda() if rot else None
    hs, offs = 3 * hd, (0, hd, 2 * hd)
    scale = hd ** -0.5
    k2, v2 = kc.clone(), vc.clone()
    flat = ops.attn_decode_step(qkv, nh, hs, hd, offs, rot, cos, sin, k2, v2, pos, scale, None)
    k1, v1 = kc.clone(), vc.clone()
    got = ops.attn_decode_step(qkv, nh, hs, hd, offs, rot, cos, sin, k1, v1, pos, scale, None, K, shared, group_mode=1)
    torch.cuda.synchronize()
    assert torch.equal(k1, k2) and torch.equal(v1, v2), "cache after the step"
    close(got, flat.float(), rel=2 ** -7, name="grouped vs ungrouped")
    # groups by ADDRESS: the prompt's keys of every row come from the group's first row -- poison the other rows' copies to prove it; the bits of the ungrouped form
    k3, v3 = kc.clone(), vc.clone()
    for gi in range(groups):
        n = int(shared[gi])
        k3[gi * K + 1:(gi + 1) * K, :n] = float("nan")
        v3[gi * K + 1:(gi + 1) * K, :n] = float("nan")
    byaddr = ops.attn_decode_step(qkv, nh, hs, hd, offs, rot, cos, sin, k3, v3, pos, scale, None, K, shared, group_mode=0)
    assert torch.equal(byaddr, flat), "groups by address: same keys, same partition, same bits as ungrouped"
    q2 = qkv.clone()
    ops.decode_rope_append(q2, nh, hs, hd, offs, rot, cos, sin, kc.clone(), vc.clone(), pos)
    qf, kf, vf = q2.view(R, nh, 3 * hd)[..., :hd].float(), k1.float(), v1.float()
    for r in range(R):
        n = int(pos[r]) + 1
        sc = torch.einsum("hd,khd->hk", qf[r], kf[r, :n]) * scale
        close(got[r, 0], torch.einsum("hk,khd->hd", torch.softmax(sc, -1), vf[r, :n]), name=f"row {r} vs fp32")
    for _ in range(30):
        assert torch.equal(got, ops.attn_decode_step(qkv, nh, hs, hd, offs, rot, cos, sin, k1, v1, pos, scale, None, K, shared, group_mode=1)), "two launches differ"


@pytest.mark.parametrize("K,groups,nh,hd", [(10, 1, 32, 80), (5, 3, 8, 64), (3, 2, 12, 64), (16, 1, 4, 128)])
def test_kv_reorder_beams(ops, K, groups, nh, hd):
    """unimp_kv_reorder_beams against transformers' _reorder_cache as decode.py ran it before (tail.copy_(tail.index_select(2, local)) per
    group): the generated slots [slot0, pos) of every layer / K / V plane follow their hypotheses, bit for bit; the prompt's slots, the slots
    the decode has not reached and the other groups' rows are untouched; groups sit at different lengths."""
    layers, cap, max_new = 3, 48, 12
    g = torch.Generator().manual_seed(K * 100 + groups)
    R = K * groups
    kv = torch.randn(layers, 2, R, cap, nh, hd, generator=g).to(bf16).cuda()
    slot0 = torch.tensor([20 + 3 * i for i in range(groups)], dtype=torch.int32)
    done = torch.tensor([7, 0, 12][:groups] if groups <= 3 else [5] * groups)             # generated so far per group (0: nothing to move; 12: the whole tail)
    pos = (slot0.long() + done).repeat_interleave(K).cuda()
    local = torch.stack([torch.randint(0, K, (K,), generator=g) for _ in range(groups)]).view(-1).cuda()
    want = kv.clone()
    for b in range(groups):
        L0, n = int(slot0[b]), int(done[b])
        tail = want[:, :, b * K:(b + 1) * K, L0:L0 + n]
        tail.copy_(tail.index_select(2, local[b * K:(b + 1) * K]))
    ops.kv_reorder_beams(kv, K, local, slot0.cuda(), pos, max_new)
    torch.cuda.synchronize()
    assert torch.equal(kv, want)


def test_gemm_skinny_rejects_unsupported(ops):
    a, b = rnd(8, 72, seed=1).cuda(), rnd(16, 72, seed=2).cuda()
    with pytest.raises(Exception):
        ops.gemm(a, b, variant="skinny")          # K % 64 != 0
    a, b = rnd(128, 64, seed=1).cuda(), rnd(16, 64, seed=2).cuda()
    with pytest.raises(Exception):
        ops.gemm(a, b, variant="skinny")          # M > 64


@pytest.mark.parametrize("M", [96, 1024])
def test_gemm_padded_vocab_like(ops, M):
    """lm-head shape class: N odd (74053-like), padded ldc; dX/dW read the padded dlogits."""
    V, H = 1005, 128
    ldv = 1008
    h, w = rnd(M, H, seed=10), rnd(V, H, seed=11, scale=0.3)
    logits = ops.gemm(h.cuda(), w.cuda(), ldc=ldv)
    assert logits.shape == (M, V) and logits.stride(0) == ldv
    close(logits, h.float() @ w.float().t(), name="head fwd")
    dl = torch.zeros(M, ldv, dtype=bf16)
    dl[:, :V] = rnd(M, V, seed=12)
    dld = dl.cuda()
    dh = ops.gemm(dld[:, :V], w.cuda(), b_ks=True)                   # [M,V] @ [V,H]
    close(dh, dl[:, :V].float() @ w.float(), name="head dX")
    dw = ops.gemm(dld[:, :V], h.cuda(), a_ks=True, b_ks=True)        # dl^T @ h -> [V,H]
    close(dw, dl[:, :V].float().t() @ h.float(), name="head dW")


@pytest.mark.parametrize("M,N,K,gate", [(512, 1024, 4096, False), (256, 264, 8192 + 72, True), (1024, 512, 2048, False),
                                        # 256 x 256 ping-pong tiles under split-K: ragged M / N / K, kc and ks operand forms
                                        (1024, 4096, 6144, False), (776, 1032, 4096 + 40, True), (128, 2560, 4096, False)])
def test_gemm_splitk_weight_grad(ops, M, N, K, gate):
    """dW-shaped problem (both operands k-strided, tiny output, deep K): the split-K path is picked automatically."""
    dy, x = rnd(K, M, seed=1, scale=0.1), rnd(K, N, seed=2)
    g = torch.tensor([0.4]).to(bf16) if gate else None
    want = dy.float().t() @ x.float() * (math.tanh(0.4) if gate else 1.0)
    got = ops.gemm(dy.cuda(), x.cuda(), a_ks=True, b_ks=True, gate=g.cuda() if gate else None)
    close(got, want, name="splitk dW")
    ref = ops.gemm(dy.cuda(), x.cuda(), a_ks=True, b_ks=True, gate=g.cuda() if gate else None, variant="v1")
    close(ref, want, name="v1 dW")
    assert torch.equal(got, ops.gemm(dy.cuda(), x.cuda(), a_ks=True, b_ks=True, gate=g.cuda() if gate else None))   # reproducible
    if K % 8 == 0:                                   # the same product from k-contiguous operands (kc x kc split-K)
        got_kc = ops.gemm(dy.t().contiguous().cuda(), x.t().contiguous().cuda(), gate=g.cuda() if gate else None, _splits=-1)   # (a forward-form call never splits by itself)
        close(got_kc, want, name="splitk kc")
    # accumulate: the reduction pass adds into C (the weight-gradient sink of train.Trainer), bf16 and fp32 outputs
    c0 = rnd(M, N, seed=3)
    acc = c0.cuda().clone()
    ops.gemm(dy.cuda(), x.cuda(), a_ks=True, b_ks=True, gate=g.cuda() if gate else None, out=acc, accumulate=True)
    close(acc, want + c0.float(), name="splitk dW accumulate")
    acc32 = c0.float().cuda()
    ops.gemm(dy.cuda(), x.cuda(), a_ks=True, b_ks=True, gate=g.cuda() if gate else None, out=acc32, accumulate=True)
    close(acc32, want + c0.float(), name="splitk dW accumulate f32")


def test_gemm_splitk_head_dx_shape(ops):
    """The LM head's dX over the labeled rows only: [~640, V] x [V, 2560] with V = 74 053 (odd, the logit-gradient rows padded to a
    multiple of 8): k-contiguous A, k-strided B, 100 tiles of 128 x 128 and a very deep K -> split-K (was one 170 TFLOP/s launch)."""
    M, N, K = 640, 2560, 20000 + 5
    ld = (K + 7) // 8 * 8
    dl = torch.zeros(M, ld, dtype=bf16)
    dl[:, :K] = rnd(M, K, seed=1, scale=0.05)
    w = rnd(K, N, seed=2)
    got = ops.gemm(dl.cuda()[:, :K], w.cuda(), b_ks=True, _splits=-1)              # inside a backward pass this form splits by itself
    close(got, dl[:, :K].float() @ w.float(), name="head dX split-K")
    assert torch.equal(got, ops.gemm(dl.cuda()[:, :K], w.cuda(), b_ks=True, _splits=-1))
    close(ops.gemm(dl.cuda()[:, :K], w.cuda(), b_ks=True, variant="v1"), dl[:, :K].float() @ w.float(), name="head dX v1")


@pytest.mark.parametrize("M,N", [(4352, 4096), (4096, 4608)])
def test_gemm_tail_split_weight_grad(ops, M, N):
    """Weight-gradient GEMM with 257..511 tiles (one full round of the 256 CUs + a mostly idle second one): ops.gemm cuts the
    output into a <= 256-tile part and a split-K remainder (ops._tail_split_plan).  Same result as the single-launch kernel,
    reproducible, with gate and with accumulation into an existing gradient."""
    K = 8192 + 64
    plan = ops._tail_split_plan(M, N, K)
    assert plan is not None, plan
    dy, x = rnd(K, M, seed=1, scale=0.1).cuda(), rnd(K, N, seed=2).cuda()
    g = torch.tensor([0.4]).to(bf16).cuda()
    want = ops.gemm(dy, x, a_ks=True, b_ks=True, gate=g, variant="pp256")        # a forced variant bypasses the split
    got = ops.gemm(dy, x, a_ks=True, b_ks=True, gate=g)
    close(got, want.float(), name="tail-split dW")
    assert torch.equal(got, ops.gemm(dy, x, a_ks=True, b_ks=True, gate=g))
    spot = torch.randint(0, M, (64,)), torch.randint(0, N, (64,))
    ref = (dy[:, spot[0].cuda()].float() * x[:, spot[1].cuda()].float()).sum(0) * math.tanh(0.4)
    close(got[spot[0].cuda(), spot[1].cuda()], ref, name="tail-split dW vs fp32 spot check")
    c0 = rnd(M, N, seed=3).cuda()
    acc = c0.clone()
    ops.gemm(dy, x, a_ks=True, b_ks=True, gate=g, out=acc, accumulate=True)
    close(acc, want.float() + c0.float(), name="tail-split dW accumulate")


def test_gemm_rejects_bad_args(ops):
    from unimp_amd._lib import UnimpHipError
    a = rnd(16, 12).cuda()      # K = 12: ld not a multiple of 8
    with pytest.raises(UnimpHipError):
        ops.gemm(a, a)
    with pytest.raises(UnimpHipError):
        ops.gemm(rnd(8, 8), rnd(8, 8))       # CPU tensors: no fallback


# ------------------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("rows,D", [(5, 64), (300, 1024), (257, 768), (130, 2560), (33, 4096)])
@pytest.mark.parametrize("rms", [False, True])
def test_layernorm_fwd_bwd(ops, rows, D, rms):
    x, g, b, dy = rnd(rows, D, seed=1, scale=2.0), (1 + 0.3 * rnd(D, seed=2).float()).to(bf16), rnd(D, seed=3), rnd(rows, D, seed=4)
    eps = 1e-5
    xr = x.float().requires_grad_(True)
    gr, br = g.float().requires_grad_(True), b.float().requires_grad_(True)
    if rms:
        y = gr * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + eps))
    else:
        y = torch.nn.functional.layer_norm(xr, (D,), gr, br, eps)
    y.backward(dy.float())
    out, mean, rstd = ops.layernorm_fwd(x.cuda(), g.cuda(), None if rms else b.cuda(), eps, rms=rms)
    close(out, y.detach(), name="ln fwd")
    res = rnd(rows, D, seed=5)
    dx, dg, db = ops.layernorm_bwd(dy.cuda(), x.cuda(), g.cuda(), mean, rstd, dres=res.cuda(), want_wgrad=True,
                                   has_beta=not rms, rms=rms)
    close(dx, xr.grad + res.float(), name="ln dx+res")
    close(dg, gr.grad, rel=2 ** -6, name="ln dgamma")
    if not rms:
        close(db, br.grad, rel=2 ** -6, name="ln dbeta")
    dx2, _, _ = ops.layernorm_bwd(dy.cuda(), x.cuda(), g.cuda(), mean, rstd, rms=rms)
    close(dx2, xr.grad, name="ln dx")


def test_layernorm_grouped_rows(ops):
    """Perceiver cat(x, latents): two LNs write into one [G, 5, D] buffer."""
    G, D = 7, 128
    xa, xb = rnd(G * 3, D, seed=1), rnd(G * 2, D, seed=2)
    g, b = rnd(D, seed=3), rnd(D, seed=4)
    buf = torch.zeros(G * 5, D, dtype=bf16, device="cuda")
    ops.layernorm_fwd(xa.cuda(), g.cuda(), b.cuda(), 1e-5, out=buf, grp=3, grp_stride=5, grp_off=0)
    ops.layernorm_fwd(xb.cuda(), g.cuda(), b.cuda(), 1e-5, out=buf, grp=2, grp_stride=5, grp_off=3)
    ya = torch.nn.functional.layer_norm(xa.float(), (D,), g.float(), b.float())
    yb = torch.nn.functional.layer_norm(xb.float(), (D,), g.float(), b.float())
    want = torch.cat([ya.view(G, 3, D), yb.view(G, 2, D)], 1).reshape(G * 5, D)
    close(buf, want, name="grouped ln")


# ------------------------------------------------------------------------------------------------- attention
def attn_ref(q, k, v, scale, mode, kv_len=None, seg=None, seg_len=0):
    """q [B,Sq,H,D] etc, fp32 reference with explicit masks; fully-masked rows -> 0."""
    B, Sq, H, D = q.shape
    Sk = k.shape[1]
    s = torch.einsum("bqhd,bkhd->bhqk", q, k) * scale
    ok = torch.ones(B, 1, Sq, Sk, dtype=torch.bool)
    kk = torch.arange(Sk)[None, None, None, :]
    if kv_len is not None:
        ok = ok & (kk < kv_len[:, None, None, None])
    if mode == 1:
        ok = ok & (kk <= torch.arange(Sq)[None, None, :, None])
    if mode == 2:
        t = seg[:, None, :, None]
        ok = ok & (kk >= (t - 1) * seg_len) & (kk < t * seg_len) & (t > 0)
    s = s.masked_fill(~ok, float("-inf"))
    p = torch.softmax(s, -1)
    p = torch.nan_to_num(p, nan=0.0)
    return torch.einsum("bhqk,bkhd->bqhd", p, v)


ATTN_CASES = [
    # B, H, Sq, Sk, D, mode
    (2, 3, 64, 64, 64, 0), (2, 2, 257, 257, 64, 0), (1, 2, 64, 320, 64, 0), (2, 4, 200, 200, 80, 1),
    (1, 2, 512, 512, 80, 1), (2, 2, 130, 130, 128, 1), (2, 8, 100, 192, 64, 2), (1, 2, 96, 96, 64, 1),
    (2, 2, 1024, 1024, 128, 1),      # cfg5's workload: MPT head dim 128 over an image-generation sequence (L = 1024)
    (1, 4, 1000, 1000, 80, 1), (1, 2, 257, 257, 80, 0), (1, 8, 512, 1024, 64, 2),     # cfg4: 16 images x 64 latents
    (1, 3, 257, 257, 64, 0),         # the ViT-L/14 form (B = 1: no kv_len): last key seeds the softmax state, 9 waves per block
    # forms attention3.hip's dK/dV kernel serves (generation 2): 1, 2 and 3 key super-blocks of 256, kv_len inside a block, a ragged last super-block
    (2, 2, 512, 512, 80, 1), (2, 3, 256, 256, 80, 0), (1, 2, 96, 96, 80, 1), (2, 2, 640, 640, 80, 1), (1, 2, 64, 320, 80, 0), (2, 1, 32, 32, 80, 1),
]


def test_attention3_dispatch_threshold_and_data_parallel_switch(ops):
    """which dK/dV kernel serves the LM's form: attention3.hip from one (batch, head) pair per CU on, the first generation below it, and
    the first generation as well while ops.AVOID_PERSISTENT is set (a data-parallel group's collectives share the CUs: no persistent
    kernels) -- with the same gradients either way."""
    from unimp_amd import _lib
    L = _lib.lib()
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    D, S, H = 80, 64, 8
    assert L.unimp_attn_get_generation() == 2
    def run(B):
        g = torch.Generator().manual_seed(B)
        qkv = torch.randn(B, S, H, 3 * D, generator=g).to(bf16).cuda()
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        do = torch.randn(B, S, H, D, generator=g).to(bf16).cuda()
        o, lse = ops.attn_fwd(q, k, v, D ** -0.5, ops.MASK_CAUSAL)
        dqkv = torch.zeros_like(qkv)
        ops.attn_bwd(q, k, v, o, lse, do, dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:], D ** -0.5, ops.MASK_CAUSAL)
        torch.cuda.synchronize()
        return dqkv.float().cpu()
    big, small = (ncu + H - 1) // H, max(1, (ncu - 1) // H)
    a = run(big)
    assert L.unimp_attn_last_dkv() == 3
    run(small)
    assert L.unimp_attn_last_dkv() == 1
    was, ops.AVOID_PERSISTENT = ops.AVOID_PERSISTENT, True
    try:
        b = run(big)
        assert L.unimp_attn_last_dkv() == 1
    finally:
        ops.AVOID_PERSISTENT = was
    close(a[..., D:], b[..., D:], rel=2 ** -5, name="dk / dv: attention3 vs first generation")


@pytest.mark.parametrize("B,H,S,mode", [(40, 8, 64, 1), (35, 8, 320, 1), (33, 8, 288, 0), (48, 6, 512, 1)])
def test_attention3_many_items_per_workgroup(ops, B, H, S, mode):
    """attention3.hip's persistent workgroups with MORE (batch, head) pairs than CUs: every workgroup walks several items, the next
    item's K / V rows and first tiles ride in the current item's loop.  Short sequences (fewer tiles than the pipeline is deep: the
    after-loop fetches), two super-blocks with a short second one, key lengths that leave a whole super-block without a visible key
    (an item of zero tiles), against the first-generation kernel and the fp32 reference."""
    from unimp_amd import _lib
    D = 80
    g = torch.Generator().manual_seed(S + B)
    qkv = torch.randn(B, S, H, 3 * D, generator=g).to(bf16).cuda()
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    do = torch.randn(B, S, H, D, generator=g).to(bf16).cuda()
    kv_len = torch.randint(1, S + 1, (B,), generator=g).to(torch.int32)
    kv_len[0] = S
    kv_len[1] = min(S, 37)
    kv_len = kv_len.cuda()
    res = {}
    old3 = _lib.lib().unimp_attn_set_dkv3(2)
    try:
        for gen in (4, 2):
            old = _lib.lib().unimp_attn_set_generation(gen)
            try:
                o, lse = ops.attn_fwd(q, k, v, D ** -0.5, mode, kv_len)
                dqkv = torch.full_like(qkv, float("nan"))
                dq, dk, dv = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
                ops.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, D ** -0.5, mode, kv_len)
                torch.cuda.synchronize()
                res[gen] = (dk.float().cpu(), dv.float().cpu())
            finally:
                _lib.lib().unimp_attn_set_generation(old)
    finally:
        _lib.lib().unimp_attn_set_dkv3(old3)
    # two implementations of one contract: same values up to the bf16 rounding of P / dS and of the results
    close(res[2][1], res[4][1], rel=2 ** -6, name="attention3 dv vs first generation")
    close(res[2][0], res[4][0], rel=2 ** -5, name="attention3 dk vs first generation")
    # and the first batch rows against the fp32 reference
    nb = 2
    qr, kr, vr = (t[:nb].float().cpu().requires_grad_(True) for t in (q, k, v))
    want = attn_ref(qr, kr, vr, D ** -0.5, mode, kv_len[:nb].long().cpu(), None, 0)
    want.backward(do[:nb].float().cpu())
    close(res[2][0][:nb], kr.grad, rel=2 ** -5, name="attention3 dk")
    close(res[2][1][:nb], vr.grad, rel=2 ** -5, name="attention3 dv")


@pytest.mark.parametrize("B,H,Sq,Sk,D,mode", ATTN_CASES)
def test_attention_fwd_bwd(ops, attn_gen, B, H, Sq, Sk, D, mode):
    g = torch.Generator().manual_seed(Sq * 7 + D)
    # packed [B, S, H, 3D] buffer like the fused QKV GEMM output: exercises strided views
    if Sq == Sk:
        qkv = (torch.randn(B, Sq, H, 3 * D, generator=g)).to(bf16)
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    else:
        q = torch.randn(B, Sq, H, D, generator=g).to(bf16)
        kv = torch.randn(B, Sk, 2, H, D, generator=g).to(bf16)
        k, v = kv[:, :, 0], kv[:, :, 1]
    kv_len = seg = None
    seg_len = 0
    if mode in (0, 1) and Sq == Sk and B > 1:
        kv_len = torch.tensor([Sk, max(1, Sk - 37)][:B], dtype=torch.int32)
    if mode == 2:
        seg_len = 64
        T = Sk // seg_len
        seg = torch.zeros(B, Sq, dtype=torch.int32)
        for b in range(B):
            cuts = sorted(torch.randint(3, Sq, (T,), generator=g).tolist())
            for c in cuts:
                seg[b, c:] += 1
        seg.clamp_(max=T)
    scale = D ** -0.5
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    want = attn_ref(qr, kr, vr, scale, mode, kv_len.long() if kv_len is not None else None, seg.long() if seg is not None else None, seg_len)
    do = torch.randn(B, Sq, H, D, generator=g).to(bf16)
    want.backward(do.float())
    dev = lambda t: None if t is None else t.cuda()
    if Sq == Sk:
        qkv_d = qkv.cuda()
        qd, kd, vd = qkv_d[..., :D], qkv_d[..., D:2 * D], qkv_d[..., 2 * D:]
    else:
        qd, kvd = q.cuda(), kv.cuda()
        kd, vd = kvd[:, :, 0], kvd[:, :, 1]
    out, lse = ops.attn_fwd(qd, kd, vd, scale, mode, dev(kv_len), dev(seg), seg_len)
    close(out, want.detach(), rel=2 ** -6, name="attn fwd")
    if Sq == Sk:
        dqkv = torch.full_like(qkv_d, float("nan"))
        dq, dk, dv = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
    else:
        dq = torch.full_like(qd, float("nan"))
        dkv = torch.full_like(kvd, float("nan"))
        dk, dv = dkv[:, :, 0], dkv[:, :, 1]
    ops.attn_bwd(qd, kd, vd, out, lse, do.cuda(), dq, dk, dv, scale, mode, dev(kv_len), dev(seg), seg_len)
    close(dq, qr.grad, rel=2 ** -5, name="attn dq")
    close(dk, kr.grad, rel=2 ** -5, name="attn dk")
    close(dv, vr.grad, rel=2 ** -5, name="attn dv")


@pytest.mark.parametrize("B,H,S,D,rot,interleaved", [(2, 4, 200, 80, 80, True), (1, 2, 512, 80, 80, True), (2, 2, 130, 128, 128, False),
                                                     (2, 3, 96, 64, 16, True), (1, 2, 257, 64, 64, False)])
def test_attention_bwd_fused_inverse_rope(ops, attn_gen, B, H, S, D, rot, interleaved):
    """unimp_attn_bwd with rope tables: dq / dk leave the kernels rotated back.  Bit-identical to the separate pass
    (unimp_rope_halfsplit(inverse) over the stored bf16 gradients) for both layouts of the fused projection, full and partial
    rotary; dv untouched.  Generation 1 has no fused form: the binding says so (attn_rope_fusable) and the C entry refuses."""
    from unimp_amd._lib import UnimpHipError
    from oracle.lm import neox_rope_tables
    g = torch.Generator().manual_seed(S + D)
    if interleaved:
        qkv = torch.randn(B, S, H, 3 * D, generator=g).to(bf16).cuda()
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        hs, offs = 3 * D, (0, D)
    else:
        qkv = torch.randn(B, S, 3, H, D, generator=g).to(bf16).cuda()
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        hs, offs = D, (0, H * D)
    cos, sin = neox_rope_tables(S, rot, 10000.0)
    half = rot // 2
    ct, st = cos[:, :half].contiguous().cuda(), sin[:, :half].contiguous().cuda()
    scale = D ** -0.5
    out, lse = ops.attn_fwd(q, k, v, scale, 1)
    do = torch.randn(B, S, H, D, generator=g).to(bf16).cuda()

    def grads(rope):
        dqkv = torch.full_like(qkv, float("nan"))
        if interleaved:
            dq, dk, dv = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
        else:
            dq, dk, dv = dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2]
        ops.attn_bwd(q, k, v, out, lse, do, dq, dk, dv, scale, 1, rope=rope)
        return dqkv, (dq, dk, dv)
    # the table form of the fused rotation lives in the first / second generation kernels (attention3.hip serves the computed adjacent-pair
    # form only): the unrotated reference must come from the same dK/dV kernel for the bitwise comparison below
    from unimp_amd import _lib
    was3 = _lib.lib().unimp_attn_set_dkv3(0)
    try:
        two, (dq, dk, dv) = grads(None)
    finally:
        _lib.lib().unimp_attn_set_dkv3(was3)
    ops.rope_(two.view(B * S, -1), S, H, hs, rot, offs, ct, st, inverse=True)
    if attn_gen == 1:
        assert not ops.attn_rope_fusable(dq, dk, dv, half, D)
        with pytest.raises(UnimpHipError):
            grads((ct, st))
        return
    assert ops.attn_rope_fusable(dq, dk, dv, half, D)
    one, _ = grads((ct, st))
    assert not torch.isnan(one.float()).any()
    diff = (one.float() - two.float()).abs()
    # same inputs, same formula; the compiler may contract a*b - c*d differently in the two kernels: allow isolated 1-ulp cases
    bad = diff > 2 ** -7 * two.float().abs().clamp_min(1e-3)
    assert bad.float().mean() < 1e-3 and diff.max() <= 2 ** -6 * two.float().abs().max(), (bad.float().mean(), diff.max())


@pytest.mark.parametrize("H,D,L,lens,mode", [(4, 80, 200, (200, 77, 130), 1), (2, 64, 257, (257, 31, 5, 190), 0), (2, 128, 300, (129, 300), 1),
                                             (4, 64, 150, (150, 64, 97), 2), (3, 80, 512, (512, 300, 511, 1, 64), 1), (2, 64, 128, (128, 0, 50), 1)])
def test_attention_packed_rows(ops, H, D, L, lens, mode):
    """include/unimp_hip.h q_row_off / k_row_off: the sequences of a batch as row ranges of one [rows, H, D] buffer.  Same kernels,
    same per-sequence tiles as the padded [B, L] call with kv_len -- forward output, lse and all three gradients must come back with
    the SAME BITS on every valid row, and no row outside a sequence may be written (the buffers carry guard rows of NaN).
    mode 2: the gated cross-attention's form -- packed queries (and packed media_time), padded media keys."""
    B = len(lens)
    g = torch.Generator().manual_seed(L + D + B)
    scale = D ** -0.5
    lens_t = torch.tensor(lens, dtype=torch.int32)
    off = torch.cumsum(lens_t, 0) - lens_t
    n = int(lens_t.sum())
    M = n + 7                                                           # spare rows behind the last sequence
    rows_of = torch.cat([b * L + torch.arange(lens[b]) for b in range(B)])          # flat padded position of every packed row
    pr = ops.PackedRows(B, L, off.to(torch.int32).cuda(), lens_t.cuda(), n)
    nan = float("nan")
    if mode == 2:
        Sk, seg_len = 192, 64
        q = torch.randn(B, L, H, D, generator=g).to(bf16)
        kv = torch.randn(B, Sk, 2, H, D, generator=g).to(bf16).cuda()
        k, v = kv[:, :, 0], kv[:, :, 1]
        seg = torch.zeros(B, L, dtype=torch.int32)
        for b in range(B):
            for c in sorted(torch.randint(1, max(2, lens[b]), (Sk // seg_len,), generator=g).tolist()):
                seg[b, c:] += 1
        seg.clamp_(max=Sk // seg_len)
        do = torch.randn(B, L, H, D, generator=g).to(bf16)
        do[torch.arange(L)[None] >= lens_t[:, None]] = 0
        qd, dod, segd = q.cuda(), do.cuda(), seg.cuda()
        o_pad, lse_pad = ops.attn_fwd(qd, k, v, scale, 2, None, segd, seg_len)
        dq_pad, dkv_pad = torch.empty_like(qd), torch.empty_like(kv)
        ops.attn_bwd(qd, k, v, o_pad, lse_pad, dod, dq_pad, dkv_pad[:, :, 0], dkv_pad[:, :, 1], scale, 2, None, segd, seg_len)
        # packed queries
        qp = torch.full((1, M, H, D), 0.5, dtype=bf16); qp[0, :n] = q.view(B * L, H, D)[rows_of]
        dop = torch.zeros((1, M, H, D), dtype=bf16); dop[0, :n] = do.view(B * L, H, D)[rows_of]
        segp = torch.zeros(M, dtype=torch.int32); segp[:n] = seg.view(-1)[rows_of]
        qp, dop, segp = qp.cuda(), dop.cuda(), segp.cuda()
        out = torch.full((1, M, H, D), nan, dtype=bf16, device="cuda")
        o_p, lse_p = ops.attn_fwd(qp, k, v, scale, 2, None, segp, seg_len, out=out, q_rows=pr)
        dq_p = torch.full((1, M, H, D), nan, dtype=bf16, device="cuda")
        dkv_p = torch.empty_like(kv)
        ops.attn_bwd(qp, k, v, o_p, lse_p, dop, dq_p, dkv_p[:, :, 0], dkv_p[:, :, 1], scale, 2, None, segp, seg_len, q_rows=pr)
        assert torch.equal(o_p[0, :n], o_pad.view(B * L, H, D)[rows_of.cuda()]) and torch.isnan(o_p[0, n:].float()).all()
        assert torch.equal(dq_p[0, :n], dq_pad.view(B * L, H, D)[rows_of.cuda()]) and torch.isnan(dq_p[0, n:].float()).all()
        assert torch.equal(dkv_p, dkv_pad)
        return
    qkv = torch.randn(B, L, H, 3 * D, generator=g).to(bf16)
    do = torch.randn(B, L, H, D, generator=g).to(bf16)
    do[torch.arange(L)[None] >= lens_t[:, None]] = 0                    # what autograd delivers at <PAD> rows
    qkv_d, dod, kvl = qkv.cuda(), do.cuda(), lens_t.cuda()
    q, k, v = qkv_d[..., :D], qkv_d[..., D:2 * D], qkv_d[..., 2 * D:]
    o_pad, lse_pad = ops.attn_fwd(q, k, v, scale, mode, kvl)
    dqkv_pad = torch.zeros_like(qkv_d)
    ops.attn_bwd(q, k, v, o_pad, lse_pad, dod, dqkv_pad[..., :D], dqkv_pad[..., D:2 * D], dqkv_pad[..., 2 * D:], scale, mode, kvl)
    # the same sequences as row ranges
    qkv_p = torch.full((1, M, H, 3 * D), 0.25, dtype=bf16); qkv_p[0, :n] = qkv.view(B * L, H, 3 * D)[rows_of]
    do_p = torch.zeros((1, M, H, D), dtype=bf16); do_p[0, :n] = do.view(B * L, H, D)[rows_of]
    qkv_p, do_p = qkv_p.cuda(), do_p.cuda()
    qp, kp, vp = qkv_p[..., :D], qkv_p[..., D:2 * D], qkv_p[..., 2 * D:]
    out = torch.full((1, M, H, D), nan, dtype=bf16, device="cuda")
    o_p, lse_p = ops.attn_fwd(qp, kp, vp, scale, mode, out=out, q_rows=pr, k_rows=pr)
    dqkv_p = torch.full_like(qkv_p, nan)
    ops.attn_bwd(qp, kp, vp, o_p, lse_p, do_p, dqkv_p[..., :D], dqkv_p[..., D:2 * D], dqkv_p[..., 2 * D:], scale, mode, q_rows=pr, k_rows=pr)
    rd = rows_of.cuda()
    assert torch.equal(o_p[0, :n], o_pad.view(B * L, H, D)[rd]), float((o_p[0, :n].float() - o_pad.view(B * L, H, D)[rd].float()).abs().max())
    assert torch.isnan(o_p[0, n:].float()).all(), "the forward wrote a row outside every sequence"
    for b in range(B):
        assert torch.equal(lse_p[b, :, :lens[b]], lse_pad[b, :, :lens[b]])
    got, want = dqkv_p[0, :n], dqkv_pad.view(B * L, H, 3 * D)[rd]
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        assert torch.equal(got[..., sl], want[..., sl]), (name, float((got[..., sl].float() - want[..., sl].float()).abs().max()))
    assert torch.isnan(dqkv_p[0, n:].float()).all(), "the backward wrote a row outside every sequence"


def test_attention_spiked_row_online_softmax(ops, attn_gen):
    """force the running-max rescale branch: one key far above the rest in a late tile (cdna guide rule 26)."""
    B, H, S, D = 1, 1, 256, 64
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(B, S, H, D, generator=g).to(bf16) for _ in range(3))
    k[0, 200, 0] = q[0, 5, 0] * 6.0
    want = attn_ref(q.float(), k.float(), v.float(), D ** -0.5, 0)
    out, _ = ops.attn_fwd(q.cuda(), k.cuda(), v.cuda(), D ** -0.5, 0)
    close(out, want, rel=2 ** -6, name="spiked")


# ------------------------------------------------------------------------------------------------- RoPE
@pytest.mark.parametrize("hd,rot", [(80, 80), (64, 64), (64, 32), (96, 24)])
def test_rope(ops, hd, rot):
    from oracle.lm import neox_rope_tables, rotate_half
    B, L, nh = 2, 37, 3
    qkv = rnd(B * L, nh * 3 * hd, seed=1)
    cos, sin = neox_rope_tables(L, rot, 10000.0)
    x = qkv.float().view(B, L, nh, 3 * hd)
    want = x.clone()
    for off in (0, hd):
        r = x[..., off:off + rot]
        want[..., off:off + rot] = r * cos[None, :, None] + rotate_half(r) * sin[None, :, None]
    d = qkv.cuda().clone()
    half = rot // 2
    ct, st = cos[:, :half].contiguous().cuda(), sin[:, :half].contiguous().cuda()
    ops.rope_(d, L, nh, 3 * hd, rot, (0, hd), ct, st)
    close(d, want.view(B * L, -1), name="rope")
    ops.rope_(d, L, nh, 3 * hd, rot, (0, hd), ct, st, inverse=True)
    close(d, qkv.float(), rel=2 ** -6, name="rope inverse")


def _adjacent_perm(hd, rot):
    """new -> old index inside one head vector for the pair-adjacent rotary layout (functional._rope_perm_index)"""
    p = torch.arange(hd)
    g, j = p // 8, p % 8
    d = torch.where(j < 4, 4 * g + j, rot // 2 + 4 * g + (j - 4))
    return torch.where(p < rot, d, p)


@pytest.mark.parametrize("variant", ["pp256", "pp256p", "pp256x", "pp256px", "pp256a", "w4x"])
@pytest.mark.parametrize("nh,hd,rot,interleaved,L", [(4, 80, 80, True, 200), (6, 64, 16, True, 128), (3, 128, 128, False, 300),
                                                     (4, 80, 80, True, 2304), (2, 128, 128, False, 4100)])      # positions >= 2048
def test_gemm_rotary_epilogue(ops, variant, nh, hd, rot, interleaved, L):
    """QKV projection with the rotary epilogue (rows of W permuted to the pair-adjacent order, cos / sin computed in the
    epilogue) = plain projection + half-split RoPE from fp32 tables, dims permuted: q, k rotated, v untouched, every position."""
    from unimp_amd import functional as Fn
    from oracle.lm import neox_rope_tables, rotate_half
    B, H = 3, nh * hd
    M, K = B * L, 256
    x, w, b = rnd(M, K, seed=1), rnd(3 * H, K, seed=2, scale=0.08), rnd(3 * H, seed=3)
    idx = Fn._rope_perm_index(nh, hd, rot, interleaved, "cpu")
    rope = dict(rot=rot, hd=hd, period=3 * hd if interleaved else 3 * H, span=2 * hd if interleaved else 2 * H, L=L, log2_base=math.log2(10000.0))
    got = ops.gemm(x.cuda(), w[idx].contiguous().cuda(), bias=b[idx].contiguous().cuda(), rope=rope, variant=variant).float().cpu()
    # the same projection read from the transposed copy of the (frozen) weight -- what the training step does since round 4: same bits
    got_t = ops.gemm(x.cuda(), w[idx].t().contiguous().cuda(), b_ks=True, bias=b[idx].contiguous().cuda(), rope=rope, variant=variant).float().cpu()
    assert torch.equal(got_t, got), f"W^T form differs in {int((got_t != got).sum())} elements"
    y = (x.float() @ w.float().t() + b.float())
    y = y.view(B, L, nh, 3, hd) if interleaved else y.view(B, L, 3, nh, hd).permute(0, 1, 3, 2, 4)      # [B, L, nh, 3, hd]
    cos, sin = neox_rope_tables(L, rot, 10000.0)
    want = y.clone()
    for part in (0, 1):
        r = y[..., part, :rot]
        want[..., part, :rot] = r * cos[None, :, None] + rotate_half(r) * sin[None, :, None]
    perm = _adjacent_perm(hd, rot)
    want[..., 0, :] = want[..., 0, :][..., perm]
    want[..., 1, :] = want[..., 1, :][..., perm]
    g = got.view(B, L, nh, 3, hd) if interleaved else got.view(B, L, 3, nh, hd).permute(0, 1, 3, 2, 4)
    close(g, want, name="rotary epilogue")
    assert float((g[..., 2, :] - want[..., 2, :]).abs().max()) <= 2 ** -7 * float(want.abs().max())          # v: plain projection
    with pytest.raises(ops._lib.UnimpHipError):
        ops.gemm(x.cuda(), w.cuda(), rope=rope, variant="w8")


@pytest.mark.parametrize("B,H,S,D,rot", [(2, 4, 200, 80, 80), (1, 2, 512, 80, 80), (2, 3, 130, 128, 128), (2, 3, 96, 64, 16),
                                        (2, 2, 96, 80, 32)])       # head dim 80 rotated in part: attention3.hip's second rotation form (generation 2)
def test_attention_bwd_adjacent_inverse_rope(ops, attn_gen, B, H, S, D, rot):
    """adjacent-pair form of the fused inverse rotation (the layout of the GEMM's rotary epilogue): dq / dk of the plain
    backward, rotated back in fp32 with table cos / sin, against the kernels' on-the-fly version; dv untouched."""
    if attn_gen == 1:
        pytest.skip("generation 1 has no fused form")
    from oracle.lm import neox_rope_tables
    g = torch.Generator().manual_seed(S + D)
    qkv = torch.randn(B, S, H, 3 * D, generator=g).to(bf16).cuda()
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    scale = D ** -0.5
    out, lse = ops.attn_fwd(q, k, v, scale, 1)
    do = torch.randn(B, S, H, D, generator=g).to(bf16).cuda()

    def grads(rope):
        dqkv = torch.full_like(qkv, float("nan"))
        ops.attn_bwd(q, k, v, out, lse, do, dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:], scale, 1, rope=rope)
        return dqkv
    plain = grads(None).float().cpu().view(B, S, H, 3, D)
    fused = grads((rot // 2, math.log2(10000.0))).float().cpu().view(B, S, H, 3, D)
    cos, sin = neox_rope_tables(S, rot, 10000.0)
    c, s_ = cos[:, :rot // 2], sin[:, :rot // 2]                       # [S, half], frequency i
    want = plain.clone()
    for part in (0, 1):
        x = plain[..., part, :rot].reshape(B, S, H, rot // 8, 2, 4)      # chunk g: pairs (j, j + 4) at frequency 4 g + j
        x1, x2 = x[..., 0, :], x[..., 1, :]
        cc, ss = c.view(S, rot // 8, 4)[None, :, None], s_.view(S, rot // 8, 4)[None, :, None]
        want[..., part, :rot] = torch.stack([x1 * cc + x2 * ss, x2 * cc - x1 * ss], -2).reshape(B, S, H, rot)
    close(fused, want, rel=2 ** -7, name="adjacent inverse rope")
    assert torch.equal(fused[..., 2, :], plain[..., 2, :])


@pytest.mark.parametrize("rows,K,V,dt", [(10, 10, 74053, "bf16"), (5, 5, 50432, "bf16"), (12, 4, 1000, "f32"), (16, 16, 32000, "bf16"), (3, 1, 74053, "f32")])
def test_beam_topk_step(ops, rows, K, V, dt):
    """unimp_beam_topk against the torch ops it replaces in generate.beam_search -- log_softmax(logits.float()) + beam_scores, viewed [prompts, K * V],
    torch.topk(2 K, sorted): the same sorted scores to fp32 rounding of the normaliser, every returned index holding its score, no index twice,
    and -- where the top 2K hold no equal scores (fp32 logits) -- the same indices in the same order (bf16 logits tie inside the top 2K of 740 530
    values; torch.topk's order among equals is unspecified, this kernel's is the smaller flat index); a row that cannot win (beam score -1e9, as in the first step), ties (equal logits in two rows: the smaller flat index first), bf16
    and fp32 logits, several prompts per call; two launches give the same bits."""
    g = torch.Generator().manual_seed(rows * 7 + V)
    logits = (torch.randn(rows, V, generator=g) * 3).to(bf16 if dt == "bf16" else torch.float32).cuda()
    beam = (torch.randn(rows, generator=g) * 2).cuda()
    if K > 1:
        beam[1] = -1e9
    C = 2 * K
    assert ops.beam_topk_ok(logits, K, C)
    got_s, got_i = ops.beam_topk(logits, beam, K, C)
    scores = (torch.log_softmax(logits.float(), -1) + beam.view(-1, 1)).view(rows // K, K * V)
    want_s, want_i = torch.topk(scores, C, dim=1, largest=True, sorted=True)
    tol = 2e-6 * want_s.abs().max().item() + 1e-6
    assert (got_s - want_s).abs().max().item() <= tol
    assert (scores.gather(1, got_i) - got_s).abs().max().item() <= tol, "an index does not hold its score"
    assert all(len(set(r)) == C for r in got_i.tolist()), "an index twice"
    if all(len(set(r)) == C for r in want_s.tolist()):            # no equal scores among the winners: the order is determined
        assert torch.equal(got_i, want_i), (got_i, want_i)
    again_s, again_i = ops.beam_topk(logits, beam, K, C)
    assert torch.equal(again_s, got_s) and torch.equal(again_i, got_i)
    # ties: two rows with identical logits and beam scores -> the smaller flat index first
    if K >= 2:
        l2 = logits.clone(); l2[0] = l2[K - 1]
        b2 = beam.clone(); b2[0] = b2[K - 1] = 100.0          # the two equal rows lead
        s2, i2 = ops.beam_topk(l2, b2, K, C)
        first = i2[0].tolist()
        assert first[0] < first[1] and first[1] - first[0] == (K - 1) * V and float(s2[0, 0]) == float(s2[0, 1])


# ------------------------------------------------------------------------------------------------- embedding / misc
def test_embedding_fwd_bwd(ops):
    V, D, n = 50, 64, 300
    W, P = rnd(V, D, seed=1), rnd(20, D, seed=2)
    ids = torch.randint(0, V, (n,))
    pos = torch.randint(0, 20, (n,))
    out = ops.embedding_fwd(ids.cuda(), W.cuda())
    assert torch.equal(out.cpu(), W[ids])
    out = ops.embedding_fwd(ids.cuda(), W.cuda(), pos.cuda(), P.cuda())
    close(out, W[ids].float() + P[pos].float(), name="emb+pos")
    dout = rnd(n, D, seed=3)
    dW = ops.embedding_bwd(ids.cuda(), dout.cuda(), V)
    want = torch.zeros(V, D).index_add_(0, ids, dout.float())
    close(dW, want, name="emb bwd")


def test_embedding_bwd_is_bit_reproducible(ops):
    """the embedding gradient sums the rows of an id in a FIXED order with one writer per element (ids sorted, stable; row order inside segments of 64
    sorted positions, a run's segments in order): a hot id (pad / <image>: 60 % of the rows here, 300 segments) whose rows span five orders of
    magnitude gives the same bits launch after launch and the fp32 sum of that very order; ids that fill exactly one / two segments, runs that end on
    a segment boundary; out-of-range ids are skipped.  (The fp32-atomic form this replaces differed in 3 of 20 launches at the cfg2 shape.)"""
    rows, D, V = 32768, 256, 5000
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, V, (rows,), generator=g)
    ids[torch.rand(rows, generator=g) < 0.6] = 7
    ids[5], ids[9] = -1, V + 3
    ids[ids == 11] = 12; ids[100:164] = 11                      # id 11: exactly 64 rows; id 13: exactly 128
    ids[ids == 13] = 12; ids[200:328] = 13
    dout = (torch.randn(rows, D, generator=g) * torch.logspace(-3, 1, rows)[:, None]).to(bf16)
    idc, dc = ids.cuda(), dout.cuda()
    base = ops.embedding_bwd(idc, dc, V)
    for _ in range(20):
        assert torch.equal(base, ops.embedding_bwd(idc, dc, V))
    sid, perm = torch.sort(ids, stable=True)                     # the kernels' order, restated: segments of 64 sorted positions
    pos7 = torch.nonzero(sid == 7).flatten().tolist()
    hot, seg, cur = torch.zeros(D), None, None
    for t in pos7:
        if seg != t // 64:
            if cur is not None:
                hot += cur
            seg, cur = t // 64, torch.zeros(D)
        cur += dout[perm[t]].float()
    hot += cur
    assert torch.equal(base[7].cpu(), hot.to(bf16))
    ok = (ids >= 0) & (ids < V)
    want = torch.zeros(V, D, dtype=torch.float64).index_add_(0, ids[ok], dout[ok].double())
    close(base, want.float(), name="emb bwd, hot id")


def test_vit_patchify_assemble(ops):
    N, P, Hi = 3, 14, 56
    px = torch.randn(N, 3, Hi, Hi)
    K = 3 * P * P
    ld = (K + 7) // 8 * 8
    cols = ops.vit_patchify(px.cuda(), P, ld)
    want = torch.nn.functional.unfold(px, P, stride=P).transpose(1, 2).reshape(-1, K)   # (N*g*g, c*P*P+py*P+px)
    close(cols[:, :K], want, name="patchify")
    assert (cols[:, K:] == 0).all()
    D, g2 = 64, (Hi // P) ** 2
    patch, cls, pos = rnd(N * g2, D, seed=1), rnd(D, seed=2), rnd(g2 + 1, D, seed=3)
    x = ops.vit_assemble(patch.cuda(), cls.cuda(), pos.cuda(), N, g2)
    want = torch.cat([cls.float().expand(N, 1, D), patch.float().view(N, g2, D)], 1) + pos.float()
    close(x, want, name="assemble")


def test_small_elementwise(ops):
    a, b = rnd(1000, 24, seed=1), rnd(1000, 24, seed=2)
    close(ops.add(a.cuda(), b.cuda()), a.float() + b.float(), name="add")
    close(ops.dot(a.cuda(), b.cuda()), (a.float() * b.float()).sum()[None], rel=1e-4, name="dot")
    f = torch.randn(1003)
    close(ops.cast_bf16(f.cuda(), 0.5), (f * 0.5), name="cast")
    src = rnd(4, 64, seed=3)
    out = ops.bcast_rows(src.cuda(), 20)
    assert torch.equal(out.cpu(), src.repeat(5, 1))
    big = rnd(20, 64, seed=4)
    close(ops.reduce_rows_periodic(big.cuda(), 4), big.float().view(5, 4, 64).sum(0), name="reduce periodic")
    gu = rnd(50, 2 * 40, seed=5)
    close(ops.swiglu_fwd(gu.cuda(), 40), torch.nn.functional.silu(gu.float()[:, :40]) * gu.float()[:, 40:], name="swiglu")
    x = gu.float().requires_grad_(True)
    dout = rnd(50, 40, seed=6)
    (torch.nn.functional.silu(x[:, :40]) * x[:, 40:]).backward(dout.float())
    close(ops.swiglu_bwd(gu.cuda(), dout.cuda(), 40), x.grad, name="swiglu bwd")


# ------------------------------------------------------------------------------------------------- train step kernels
def test_label_mask_bit_exact(ops):
    from oracle import train_step as ts
    rng = np.random.default_rng(0)
    for L in (24, 64, 130, 512):
        ids = rng.integers(0, 12, size=(5, L))
        want = ts.label_mask_loop(ids, 8, 9, 10, 11)
        labels, mt = ops.label_mask(torch.from_numpy(ids).cuda(), 8, 9, 10, 11)
        assert np.array_equal(labels.cpu().numpy(), want)
        assert np.array_equal(mt.cpu().numpy(), np.cumsum(ids == 11, 1))


def test_label_mask_golden(ops, golden_dir):
    import os
    z = np.load(os.path.join(golden_dir, "train_step_g2_rw1.npz"))
    ANS, EOC, PAD, IMG = [int(v) for v in z["special"]]
    labels, _ = ops.label_mask(torch.from_numpy(z["ids"]).cuda(), ANS, EOC, PAD, IMG)
    assert np.array_equal(labels.cpu().numpy(), z["labels"])


@pytest.mark.parametrize("gamma,rw", [(0, 0), (0, 1), (2, 0), (2, 1)])
def test_focal_ce_golden(ops, golden_dir, gamma, rw):
    """reference-captured loss / dlogits (mmrec.train_one_epoch) with bf16-rounded logits as input."""
    import os
    from oracle import train_step as ts
    z = np.load(os.path.join(golden_dir, f"train_step_g{gamma}_rw{rw}.npz"))
    logits = torch.from_numpy(z["logits"]).to(bf16)
    labels, weights = torch.from_numpy(z["labels"]), torch.from_numpy(z["weights"])
    B, L, V = logits.shape
    ldv = (V + 7) // 8 * 8 + 8
    buf = torch.full((B, L, ldv), 7.0, dtype=bf16)
    buf[..., :V] = logits
    d = buf.cuda()
    lse, zy, out3 = ops.focal_ce_fwd(d[..., :V], V, labels.cuda(), weights.cuda(), float(gamma), bool(rw))
    want = ts.weighted_focal_ce(logits.float(), labels, weights, gamma, bool(rw))
    o = out3.cpu()
    assert abs(o[0] / o[1] - want) <= 1e-4 * abs(want), (o, want)
    assert o[1] == (labels[:, 1:] != -100).sum()
    gs = torch.ones(1, device="cuda")
    ops.focal_ce_bwd(d[..., :V], V, labels.cuda(), weights.cuda(), float(gamma), bool(rw), lse, zy, out3, gs, d[..., :V])
    wantg = ts.focal_ce_dlogits(logits.float(), labels, weights, gamma, bool(rw))
    close(d[..., :V], wantg, name="dlogits")
    assert (d[..., V:] == 0).all()
    # against the reference's own fp32 capture (inputs differ by bf16 rounding of the logits only)
    assert abs(o[0] / o[1] - float(z["loss"])) <= 5e-3 * abs(float(z["loss"]))


def test_sumsq_adamw(ops):
    from oracle import train_step as ts
    n, n_decay = 10007, 4096
    g = rnd(n, seed=1, scale=0.05)
    p0 = torch.randn(n, generator=torch.Generator().manual_seed(2))
    master, m, v = p0.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    p16 = p0.to(bf16).cuda()
    pr, mr, vr = p0.clone(), torch.zeros(n), torch.zeros(n)
    buf = torch.zeros(1025, device="cuda")
    for step in (1, 2, 3):
        gd = g.cuda().clone()
        buf[0] = 0
        ops.sumsq(gd, buf)
        tot, coef = ts.clip_coef([g.float() * 0.5], 1.0)
        assert abs(buf[0].item() ** 0.5 * 0.5 - tot) < 1e-3 * tot
        ops.adamw_flat(master, m, v, p16, gd, n_decay, 1e-2, 0.9, 0.999, 1e-8, 0.1, step, buf, 0.5, 1.0, True)
        geff = g.float() * 0.5 * coef
        ts.adamw_step(pr[:n_decay], geff[:n_decay], mr[:n_decay], vr[:n_decay], step, 1e-2, 0.1)
        ts.adamw_step(pr[n_decay:], geff[n_decay:], mr[n_decay:], vr[n_decay:], step, 1e-2, 0.0)
        assert (gd == 0).all()
    assert torch.allclose(master.cpu(), pr, rtol=1e-4, atol=1e-6)
    assert torch.allclose(m.cpu(), mr, rtol=1e-4, atol=1e-7) and torch.allclose(v.cpu(), vr, rtol=1e-4, atol=1e-9)
    assert torch.equal(p16.cpu(), master.cpu().to(bf16))


def mpt_alibi_slopes(n_heads, alibi_bias_max=8):
    """transformers' build_mpt_alibi_tensor slopes (models/mpt/modeling_mpt.py)."""
    import math
    n2 = 2 ** math.ceil(math.log2(n_heads))
    base = torch.arange(1, n2 + 1, dtype=torch.float32) * (alibi_bias_max / n2)
    slopes = 1.0 / torch.pow(2, base)
    if n2 != n_heads:
        slopes = torch.cat([slopes[1::2], slopes[::2]])[:n_heads]
    return slopes.contiguous()


@pytest.mark.parametrize("B,H,S,D,kvpad", [(2, 4, 200, 64, True), (1, 6, 512, 128, False), (2, 3, 96, 80, True),
                                           (2, 4, 1024, 128, True)])      # the last: cfg5's img-gen sequence length
def test_attention_alibi_causal(ops, attn_gen, B, H, S, D, kvpad):
    """ALiBi as MPT applies it (bias = slope_h * (j - (S - 1)), causal): forward and all three gradients, plus a single
    query row against a longer key cache (the decode shape)."""
    g = torch.Generator().manual_seed(S + D)
    qkv = torch.randn(B, S, H, 3 * D, generator=g).to(bf16)
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    slopes = mpt_alibi_slopes(H)
    kv_len = torch.tensor([S, S - 29][:B], dtype=torch.int32) if kvpad else None
    scale = D ** -0.5
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    s = torch.einsum("bqhd,bkhd->bhqk", qr, kr) * scale
    s = s + slopes[None, :, None, None] * (torch.arange(S)[None, None, None, :] - (S - 1))
    ok = torch.arange(S)[None, None, None, :] <= torch.arange(S)[None, None, :, None]
    if kv_len is not None:
        ok = ok & (torch.arange(S)[None, None, None, :] < kv_len.long()[:, None, None, None])
    want = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s.masked_fill(~ok, float("-inf")), -1), vr)
    do = torch.randn(B, S, H, D, generator=g).to(bf16)
    want.backward(do.float())
    qkv_d = qkv.cuda()
    qd, kd, vd = qkv_d[..., :D], qkv_d[..., D:2 * D], qkv_d[..., 2 * D:]
    kvl = None if kv_len is None else kv_len.cuda()
    out, lse = ops.attn_fwd(qd, kd, vd, scale, ops.MASK_CAUSAL, kvl, alibi=slopes.cuda())
    close(out, want.detach(), rel=2 ** -6, name="alibi fwd")
    dqkv = torch.full_like(qkv_d, float("nan"))
    dq, dk, dv = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
    ops.attn_bwd(qd, kd, vd, out, lse, do.cuda(), dq, dk, dv, scale, ops.MASK_CAUSAL, kvl, alibi=slopes.cuda())
    close(dq, qr.grad, rel=2 ** -5, name="alibi dq")
    close(dk, kr.grad, rel=2 ** -5, name="alibi dk")
    close(dv, vr.grad, rel=2 ** -5, name="alibi dv")
    # decode: the last query row alone against the whole key range (MASK_NONE + kv_len)
    n = S if kv_len is None else int(kv_len.min())
    o1, _ = ops.attn_fwd(qd[:, n - 1:n], kd, vd, scale, ops.MASK_NONE, torch.full((B,), n, dtype=torch.int32, device="cuda"), alibi=slopes.cuda())
    rows = torch.softmax((torch.einsum("bhd,bkhd->bhk", q[:, n - 1].float(), k[:, :n].float()) * scale
                          + slopes[None, :, None] * torch.arange(n)[None, None, :]), -1)
    close(o1[:, 0], torch.einsum("bhk,bkhd->bhd", rows, v[:, :n].float()), rel=2 ** -6, name="alibi decode row")


@pytest.mark.parametrize("B,H,cap,D,lens,alibi", [
    (10, 32, 583, 80, "ragged", False),        # eval_rec at cfg2: 10 beams, 469-token prompt + 50 new tokens (+64 capacity slack)
    (1, 32, 1100, 128, "full", True),          # eval_img_gen on the 9b tower: greedy, ALiBi, ~1000 keys
    (3, 4, 200, 64, "ragged", True), (2, 2, 33, 64, "one", False), (4, 6, 700, 128, "ragged", False), (2, 3, 64, 8, "full", False)])
def test_split_key_decode_attention(ops, B, H, cap, D, lens, alibi):
    """csrc/decode_attn.hip: one query row per (cache row, head) against the K/V cache [B, capacity, H, D] with per-row visible
    lengths -- against fp32 softmax attention over the visible keys, and against the training kernel on the same row (the path it
    replaces); slots beyond kv_len hold NaN (never read as values), the cache is a strided view (one K/V tensor for all layers),
    a second call gives the same bits (ordered merges, no atomics), and so does ONE row decoded alone out of a cache with twice the
    capacity (the key partition is fixed: a row's bits depend neither on its batch nor on the capacity)."""
    g = torch.Generator().manual_seed(cap + D + B)
    kv = torch.randn(2, B, cap, H, D, generator=g).to(bf16).cuda()
    q = torch.randn(B, 1, H, 3 * D, generator=g).to(bf16).cuda()[..., :D]              # a q slice of a fused qkv row
    n = {"full": [cap] * B, "one": [1] * B, "ragged": [max(1, cap - 37 * i * i - 5 * i) for i in range(B)]}[lens]
    kv_len = torch.tensor(n, dtype=torch.int32, device="cuda")
    k, v = kv[0], kv[1]
    for b in range(B):
        k[b, n[b]:] = float("nan")
        v[b, n[b]:] = float("nan")
    slopes = mpt_alibi_slopes(H).cuda() if alibi else None
    scale = D ** -0.5
    got = ops.attn_decode(q, k, v, scale, kv_len if lens != "full" else None, slopes)
    again = ops.attn_decode(q, k, v, scale, kv_len if lens != "full" else None, slopes)
    assert torch.equal(got, again)
    big = torch.full((2, 1, 2 * cap + 5, H, D), float("nan"), dtype=bf16, device="cuda")
    r = B - 1
    big[:, 0, :n[r]] = kv[:, r, :n[r]]
    alone = ops.attn_decode(q[r:r + 1], big[0], big[1], scale, kv_len[r:r + 1], slopes)
    assert torch.equal(alone[0], got[r]), "a row's bits depend on its batch / the cache capacity"
    want = torch.empty(B, 1, H, D)
    for b in range(B):
        s_ = torch.einsum("hd,khd->hk", q[b, 0].float().cpu(), k[b, :n[b]].float().cpu()) * scale
        if alibi:
            s_ = s_ + slopes.cpu()[:, None] * torch.arange(n[b])[None, :]
        want[b, 0] = torch.einsum("hk,khd->hd", torch.softmax(s_, -1), v[b, :n[b]].float().cpu())
    close(got, want, rel=2 ** -7, name="decode attention")
    if D in (64, 80, 128):
        ref, _ = ops.attn_fwd(q, k.nan_to_num(0.0), v.nan_to_num(0.0), scale, ops.MASK_NONE, kv_len, alibi=slopes)
        close(got, ref, rel=2 ** -6, name="decode attention vs training kernel")


@pytest.mark.parametrize("n_prompts,group,cap,H,D,alibi", [(1, 10, 583, 32, 80, False), (4, 10, 560, 8, 80, False), (2, 5, 800, 4, 128, True),
                                                          (3, 3, 200, 2, 64, False), (1, 16, 300, 2, 64, True), (2, 2, 130, 3, 8, False)])
def test_decode_attention_shared_prefix(ops, n_prompts, group, cap, H, D, alibi):
    """beam-search form of the decode kernel: the `group` rows of a prompt hold identical K / V below shared_len[prompt] (and their
    own keys above it); the grouped launch reads the prefix once per prompt.  Against fp32 attention per row and against the
    ungrouped kernel on the same cache; prompts of different lengths, tails of different lengths, NaN in every dead slot AND in the
    prefix slots of the rows the grouped kernel must not read (rows 1.. of a group: only row 0's prefix is touched)."""
    B = n_prompts * group
    g = torch.Generator().manual_seed(cap + D + group)
    kv = torch.randn(2, B, cap, H, D, generator=g).to(bf16)
    q = torch.randn(B, 1, H, D, generator=g).to(bf16).cuda()
    shared = [max(1, cap - 60 - 41 * p_) for p_ in range(n_prompts)]
    n = [min(cap, shared[b // group] + (7 * b) % 50) for b in range(B)]
    for p_ in range(n_prompts):                       # the beams of a prompt share the prompt's K / V
        kv[:, p_ * group:(p_ + 1) * group, :shared[p_]] = kv[:, p_ * group:p_ * group + 1, :shared[p_]]
    for b in range(B):
        kv[:, b, n[b]:] = float("nan")
    kv = kv.cuda()
    kv_len = torch.tensor(n, dtype=torch.int32, device="cuda")
    sl = torch.tensor(shared, dtype=torch.int32, device="cuda")
    slopes = mpt_alibi_slopes(H).cuda() if alibi else None
    scale = D ** -0.5
    plain = ops.attn_decode(q, kv[0], kv[1], scale, kv_len, slopes)
    poisoned = kv.clone()
    for b in range(B):
        if b % group:
            poisoned[:, b, :shared[b // group]] = float("nan")
    got = ops.attn_decode(q, poisoned[0], poisoned[1], scale, kv_len, slopes, group=group, shared_len=sl)
    assert torch.equal(got, ops.attn_decode(q, poisoned[0], poisoned[1], scale, kv_len, slopes, group=group, shared_len=sl))
    want = torch.empty(B, 1, H, D)
    for b in range(B):
        s_ = torch.einsum("hd,khd->hk", q[b, 0].float().cpu(), kv[0, b, :n[b]].float().cpu()) * scale
        if alibi:
            s_ = s_ + slopes.cpu()[:, None] * torch.arange(n[b])[None, :]
        want[b, 0] = torch.einsum("hk,khd->hd", torch.softmax(s_, -1), kv[1, b, :n[b]].float().cpu())
    close(got, want, rel=2 ** -7, name="grouped decode attention")
    close(got, plain, rel=2 ** -7, name="grouped vs plain decode attention")


# ------------------------------------------------------------------------------------------------- MX-fp8 (frozen towers, F4)
def _mx_reference(x):
    """the MX quantisation rule on the host: per 32 consecutive k, shared exponent floor(log2 amax) - 8 (E8M0 byte = exponent
    + 127), elements x / 2^e saturated to +-448 and rounded to e4m3 (torch's float8_e4m3fn cast rounds to nearest even)."""
    R, K = x.shape
    xb = x.float().view(R, K // 32, 32)
    amax = xb.abs().amax(-1)
    e = torch.where(amax > 0, torch.floor(torch.log2(amax.clamp_min(1e-38))), torch.full_like(amax, -127.0))
    sb = (e - 8 + 127).clamp(0, 254)
    scale = torch.pow(2.0, sb - 127.0)
    q = (xb / scale[..., None]).clamp(-448, 448).to(torch.float8_e4m3fn)
    return q.view(R, K), sb.to(torch.uint8)


@pytest.mark.parametrize("R,K", [(7, 64), (300, 2560), (64, 4096)])
def test_mx_quantize(ops, R, K):
    x = rnd(R, K, scale=3.0, seed=R)
    x[0, :32] = 0                                   # an all-zero block
    x[1, 5] = 1000.0                                # a block dominated by one large element
    got = ops.mx_quantize(x.cuda())
    q_ref, s_ref = _mx_reference(x)
    assert torch.equal(got.scales.cpu(), s_ref)
    assert torch.equal(got.q.cpu().view(torch.float8_e4m3fn).float(), q_ref.float())
    # round trip error of the format: relative to each block's amax, at most 2^-3 / 2 of the top binade
    back = got.dequantize().cpu()
    blk = x.float().view(R, K // 32, 32).abs().amax(-1).repeat_interleave(32, 1)
    assert ((back - x.float()).abs() <= blk * 2 ** -3).all()


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (200, 300, 256), (1024, 2560, 2560 // 128 * 128), (64, 8, 384), (520, 4104, 1024)])
def test_gemm_mxfp8_vs_dequantised_reference(ops, M, N, K):
    """the block-scaled MFMA path against fp32 matmul of the DEQUANTISED operands: the same products, fp32 accumulation in a
    different order -> 1e-5-level agreement (this pins the operand / scale layout found by tools/micro/mx_probe.hip)."""
    a = ops.mx_quantize(rnd(M, K, seed=1).cuda())
    b = ops.mx_quantize(rnd(N, K, scale=0.5, seed=2).cuda())
    want = a.dequantize().double().cpu() @ b.dequantize().double().cpu().t()
    got = ops.gemm_mx(a, b).float().cpu()
    err = (got.double() - want).abs().max() / want.abs().max()
    assert err <= 2 ** -8, err                       # bf16 output rounding dominates (2^-9 relative per element)
    # epilogue: bias + GELU with the derivative as a second output, then aux multiply + residual
    bias = rnd(N, seed=3).cuda()
    pre = torch.empty((M, N), dtype=bf16, device="cuda")
    y = ops.gemm_mx(a, b, bias=bias, act="gelu", pre=pre).float().cpu()
    z = (want + bias.float().cpu().double()).float()
    close(y, torch.nn.functional.gelu(z), rel=2 ** -6, name="mx gelu")
    zt = z.clone().requires_grad_(True)
    torch.nn.functional.gelu(zt).sum().backward()
    close(pre.float().cpu(), zt.grad, rel=2 ** -6, name="mx gelu'")
    aux, res = rnd(M, N, seed=4).cuda(), rnd(M, N, seed=5).cuda()
    y2 = ops.gemm_mx(a, b, aux=aux, res=res).float().cpu()
    close(y2, want.float() * aux.float().cpu() + res.float().cpu(), rel=2 ** -6, name="mx aux+res")


def test_gemm_mxfp8_ping_pong_kernel_epilogue_kinds(ops):
    """The 256 x 256 ping-pong MX kernel (>= 512 tiles) with the bf16 family's fixed epilogue kinds (PLAIN, RES, GELU + uint8 act'(z),
    x uint8 act'(z)) and its general epilogue (bf16 derivative forms): each against fp32 arithmetic on the dequantised operands, the
    GELU output bit-equal between the fixed kind and the general form (same accumulators, same activation code), the uint8 derivative
    within its quantisation step of the bf16 one, ragged M."""
    M, N, K = 4096 + 40, 8192, 512
    a = ops.mx_quantize(rnd(M, K, seed=1).cuda())
    b = ops.mx_quantize(rnd(N, K, scale=0.5, seed=2).cuda())
    want = (a.dequantize().double() @ b.dequantize().double().t()).float().cpu()
    bias, res = rnd(N, seed=3).cuda(), rnd(M, N, seed=5).cuda()
    close(ops.gemm_mx(a, b).float().cpu(), want, rel=2 ** -7, name="mx pp plain")
    close(ops.gemm_mx(a, b, bias=bias).float().cpu(), want + bias.float().cpu(), rel=2 ** -7, name="mx pp bias")
    close(ops.gemm_mx(a, b, res=res).float().cpu(), want + res.float().cpu(), rel=2 ** -7, name="mx pp res")
    close(ops.gemm_mx(a, b, bias=bias, res=res).float().cpu(), want + bias.float().cpu() + res.float().cpu(), rel=2 ** -7, name="mx pp bias+res")
    pre16 = torch.empty((M, N), dtype=bf16, device="cuda")
    pre8 = torch.empty((M, N), dtype=torch.uint8, device="cuda")
    y16 = ops.gemm_mx(a, b, act="gelu", pre=pre16)                  # general epilogue (bf16 derivative)
    y8 = ops.gemm_mx(a, b, act="gelu", pre=pre8)                    # EK_GELU2
    assert torch.equal(y16, y8)
    close(y8.float().cpu(), torch.nn.functional.gelu(want), rel=2 ** -6, name="mx pp gelu")
    d8 = (pre8.float() - 27.0) / 202.0
    assert float((d8 - pre16.float()).abs().max()) <= 0.5 / 202 + 2 ** -8, float((d8 - pre16.float()).abs().max())
    dy = ops.mx_quantize(rnd(M, K, seed=7).cuda())
    g16 = ops.gemm_mx(dy, b, aux=pre16).float().cpu()               # general
    g8 = ops.gemm_mx(dy, b, aux=pre8).float().cpu()                 # EK_AUX
    w2 = (dy.dequantize().double() @ b.dequantize().double().t()).float().cpu()
    close(g16, w2 * pre16.float().cpu(), rel=2 ** -6, name="mx pp aux bf16")
    close(g8, w2 * d8.cpu(), rel=2 ** -6, name="mx pp aux u8")


@pytest.mark.parametrize("M,N,K", [(4096 + 40, 8192, 512), (300, 256, 256)])
def test_gemm_mxfp8_fused_mx_output_equals_quantising_the_bf16_result(ops, M, N, K):
    """unimp_mx_gemm_desc.scale_c: the result leaves the GEMM as the next product's MX operand.  Must be the SAME BYTES (elements and
    scales) as mx_quantize of the bf16 result it replaces -- on the ping-pong kernel's two fused kinds (GELU + uint8 GELU', x uint8
    GELU'), on its general epilogue (bias + residual) and on the 128 x 128 kernel (the small shape), ragged M."""
    a = ops.mx_quantize(rnd(M, K, seed=1).cuda())
    b = ops.mx_quantize(rnd(N, K, scale=0.5, seed=2).cuda())
    bias, res = rnd(N, seed=3).cuda(), rnd(M, N, seed=5).cuda()

    def same(kw):
        want = ops.mx_quantize(ops.gemm_mx(a, b, **{k: (v.clone() if k == "pre" else v) for k, v in kw.items()}))
        got = ops.gemm_mx(a, b, out_mx=True, **kw)
        assert torch.equal(got.scales, want.scales), kw.keys()
        assert torch.equal(got.q, want.q), kw.keys()
    pre8 = torch.empty((M, N), dtype=torch.uint8, device="cuda")
    same(dict(act="gelu", pre=pre8))
    same(dict(bias=bias, act="gelu", pre=pre8))
    same(dict(aux=pre8))
    same(dict(bias=bias, res=res))
    same(dict())


@pytest.mark.parametrize("rows,D", [(300, 4096), (129, 2560), (70, 1024), (33, 96), (5, 1536), (40, 3072)])
def test_layernorm_fwd_mx_equals_quantising_the_bf16_rows(ops, rows, D):
    """unimp_layernorm_fwd_mx: the normalised rows leave as an MX operand -- same element bytes, same scales, same statistics as
    mx_quantize(layernorm_fwd(x)); full and ragged chunk maps (D = 4096 / 2560 / 1024 are the FULL forms), LayerNorm and RMSNorm."""
    x = rnd(rows, D, scale=2.0, seed=rows).cuda()
    x[0, :32] = 0
    g, b = (1 + 0.1 * rnd(D, seed=2)).cuda(), (0.1 * rnd(D, seed=3)).cuda()
    for rms, beta in ((False, b), (False, None), (True, None)):
        y, mean, rstd = ops.layernorm_fwd(x, g, beta, 1e-5, rms=rms)
        want = ops.mx_quantize(y)
        got, mean2, rstd2 = ops.layernorm_fwd_mx(x, g, beta, 1e-5, rms=rms)
        assert torch.equal(got.scales, want.scales) and torch.equal(got.q, want.q), (rms, beta is None)
        assert torch.equal(rstd, rstd2) and (rms or torch.equal(mean, mean2))


# ------------------------------------------------------------------------------------------------- packed-B ping-pong GEMM
@pytest.mark.parametrize("M,N,K", [(1024, 2560, 2560), (1096, 520, 1000), (2048, 10240, 2560), (1304, 2560, 10240 + 40), (1024, 264, 96)])
@pytest.mark.parametrize("b_ks", [False, True])
@pytest.mark.parametrize("a_ks", [False, True])
def test_gemm_packed_b_is_bit_identical(ops, M, N, K, b_ks, a_ks):
    """frozen weights pre-packed in MFMA fragment order (unimp_pack_b_bf16): the B operand bypasses the LDS; same k grouping
    inside the MFMAs, so the result must equal the unpacked ping-pong kernel's bit for bit -- both tile widths, every operand
    form, ragged M / N / K, with an epilogue, repeated launches (the packed fragments fly across a barrier: race screen)."""
    a = rnd(K, M, seed=1).cuda() if a_ks else rnd(M, K, seed=1).cuda()
    w = rnd(K, N, seed=2).cuda() if b_ks else rnd(N, K, seed=2).cuda()
    pk = ops.pack_b(w, b_ks)
    bias, res = rnd(N, seed=3).cuda(), rnd(M, N, seed=4).cuda()
    for var, pvar in (("pp256", "pk256"), ("pp128", "pk128")):
        if N < 256 and var == "pp256":
            continue
        want = ops.gemm(a, w, a_ks=a_ks, b_ks=b_ks, bias=bias, act="gelu", res=res, variant=var)
        for rep in range(4):
            got = ops.gemm(a, w, a_ks=a_ks, b_ks=b_ks, bias=bias, act="gelu", res=res, variant=pvar, b_pk=pk)
            assert torch.equal(got, want), (var, rep, (got.float() - want.float()).abs().max().item())
    ref = (a.float().t() if a_ks else a.float()) @ (w.float() if b_ks else w.float().t())
    close(ops.gemm(a, w, a_ks=a_ks, b_ks=b_ks, variant="pk256" if N >= 256 else "pk128", b_pk=pk), ref.cpu(), rel=2 ** -6, name="packed gemm")
