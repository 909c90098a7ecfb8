"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every declared symbol."""
import os
import re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "unimp_hip.h")).read()
    return sorted(set(re.findall(r"\b(unimp_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_header_symbol():
    from unimp_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = _lib.lib()
    syms = _header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), s
    assert set(_lib.declared_symbols()) == set(syms)
    assert L.unimp_abi_version() == _lib.ABI_VERSION == 8


def test_no_cpu_fallback():
    import torch
    from unimp_amd import ops, _lib
    a = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(_lib.UnimpHipError):
        ops.gemm(a, a)


def test_product_does_not_import_oracle():
    import subprocess, sys
    bad = subprocess.run(["grep", "-rlE", r"^\s*(from|import)\s+oracle", os.path.join(ROOT, "unimp_amd")],
                         capture_output=True, text=True).stdout.split()
    assert bad == [], bad


def test_ctypes_mirrors_have_the_c_struct_sizes():
    """the ctypes Structures of the binding (and of INTEGRATION.md's stub) against sizeof() inside the library."""
    import ctypes as C
    from unimp_amd import _lib
    from unimp_amd.data import _ImageDesc
    lib = C.CDLL(_lib._LIB_PATH) if hasattr(_lib, "_LIB_PATH") else _lib.lib()
    lib.unimp_struct_size.restype = C.c_int
    assert lib.unimp_struct_size(0) == C.sizeof(_lib.GemmDesc)
    assert lib.unimp_struct_size(1) == C.sizeof(_lib.AttnDesc)
    assert lib.unimp_struct_size(2) == C.sizeof(_ImageDesc)
    assert lib.unimp_struct_size(3) == C.sizeof(_lib.MxGemmDesc)
    assert lib.unimp_struct_size(9) == -1


def test_kernel_register_budgets():
    """Compile-time guard for the regressions that cost real time before: an accumulator array demoted to scratch
    (GEMM 632 -> 162 TFLOP/s), a kernel losing its second wave per SIMD to a few extra registers (dK/dV 19 -> 32 ms/step).
    hipcc's resource remarks for the hot kernels must stay inside their budgets."""
    import os
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "unimp_amd", "csrc")

    def remarks(name, extra=()):
        out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", *extra, f"-I{root}/include", f"-I{src}", "-c",
                              "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", os.devnull, os.path.join(src, name)],
                             capture_output=True, text=True, timeout=600).stderr
        res, cur = {}, None
        for line in out.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = res.setdefault(m.group(1), {})
            m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
            if m and cur is not None:
                cur[m.group(1).strip()] = int(m.group(2))
        return res
    vg = ("-mllvm", "-amdgpu-mfma-vgpr-form")
    g3 = remarks("gemm3.hip", vg)
    # 4 operand forms x 2 tile widths + the packed-B form (2 A layouts x 2 widths) with the run-time epilogue dispatch; fixed-kind
    # instantiations of the 256-wide tile (round 4): rotary x 2 B layouts, PLAIN / ACT / GELU2 / RES x 2 B layouts, AUX (k-strided B)
    assert len(g3) == 23
    for k, r in g3.items():
        assert r["ScratchSize"] == 0 and r["VGPRs"] + r.get("AGPRs", 0) <= 256 and r["Occupancy"] >= 2, (k, r)
    # the other two builds of the same file (Makefile: gemm3x.o, gemm3a.o): one fragment register set; whole-row staging of a k-contiguous A
    for flags, n in ((("-DG3X", "-DG3_ONESET"), 19), (("-DG3X", "-DG3_ONESET", "-DG3_AFULL"), 19)):
        gx = remarks("gemm3.hip", vg + flags)
        assert len(gx) == n, (flags, len(gx))            # no packed-B form in these builds
        for k, r in gx.items():
            # round 5: the K loop exists twice (steady state + the written-out tail), the residual kind's prefetched epilogue chunks live across both:
            # 244 registers at most (230 before the peel) -- still two waves per SIMD, nothing in scratch, and the faster kernel (profiles/r05_gemm_ab_*)
            assert r["ScratchSize"] == 0 and r["VGPRs"] + r.get("AGPRs", 0) <= 248 and r["Occupancy"] >= 2, (flags, k, r)
    vg_a = vg + ("-fno-slp-vectorize",)             # the Makefile's flags for the attention kernels
    at = remarks("attention.hip", vg_a)
    plain = {k: r for k, r in at.items() if "Li96ELi80ELb0" in k}
    assert len(plain) == 4                                      # forward, dQ, dK/dV with 16 and with 32 keys per wave
    for k, r in plain.items():
        assert r["ScratchSize"] == 0, (k, r)
        if "attn_dq" not in k:                                  # dQ holds two 32-row blocks of q, dO, dq: one wave per SIMD by design
            assert r["Occupancy"] >= 2, (k, r)
        if "attn_dkv" in k and k.endswith("Li1EEv5AttnPi"):     # the default dK/dV form lives on its third wave per SIMD
            assert r["Occupancy"] >= 3 and r["VGPRs"] + r.get("AGPRs", 0) <= 168, (k, r)
    # second-generation attention (the default forward and dQ): three waves per SIMD at head dim 80 / 64 is what the design buys
    a2 = remarks("attention2.hip", vg_a)
    hot = {k: r for k, r in a2.items() if ("fwd2_kernelILi80ELi4ELb0" in k or "fwd2_kernelILi64ELi5ELb0" in k or "dq2_kernelILi80ELb0" in k
                                           or "fwd2_kernelILi64ELi3ELb0ELb1" in k)}       # ...Li3ELb0ELb1: the ViT form (last key seeds the softmax)
    assert len(hot) == 4, sorted(a2)
    for k, r in hot.items():
        if "dq2" in k:       # 44 KiB of LDS (K-row and V-row images, two stages): three workgroups per CU, three waves per SIMD
            assert r["ScratchSize"] == 0 and r["VGPRs"] + r.get("AGPRs", 0) <= 168 and r["Occupancy"] >= 3, (k, r)
        else:
            assert r["ScratchSize"] == 0 and r["VGPRs"] + r.get("AGPRs", 0) <= 168 and r["Occupancy"] >= 3, (k, r)
    for k, r in a2.items():
        if "dkv2" not in k or "Li128" not in k:                  # the experimental dK/dV kernel at head dim 128 runs one wave per SIMD
            assert r["ScratchSize"] == 0, (k, r)
    # third-generation dK/dV (attention3.hip; no VGPR-form flag: its accumulators live in the AGPRs): one wave per SIMD by design, and NOTHING in
    # scratch -- with nothing co-resident a scratch reload is an exposed memory round trip (round 5: the epilogue's hoisted per-lane constants
    # reloaded from scratch cost 97 us of a 570 us launch until the lane id was made opaque per item)
    a3 = remarks("attention3.hip", ("-fno-slp-vectorize",))
    assert len(a3) == 3, sorted(a3)                          # one instantiation per rotation form of the dk epilogue (none / part of the row / the whole row)
    for k, r in a3.items():
        assert r["ScratchSize"] == 0 and r["VGPRs"] <= 256 and r.get("AGPRs", 0) <= 256, (k, r)
    # LayerNorm: HBM-bound, lives on waves in flight.  The forward and the plain backward at the LM width (5 chunks of 512) keep
    # three waves per SIMD (the scheduler widens gamma / beta or both passes' operands to fp32 if allowed to: 180-255 registers),
    # the weight-gradient form two (its partial sums are in LDS, not in 80 more registers); nothing spills at the widths in use
    ln = remarks("norm.hip")
    for k, r in ln.items():
        if "ILi5E" in k or "ILi2E" in k:
            assert r["ScratchSize"] == 0, (k, r)
        if "ln_fwd_kernelILi5ELb1E" in k or "ln_bwd_kernelILi5ELb0ELb1E" in k:
            assert r["Occupancy"] >= 3, (k, r)
        if "ln_bwd_kernelILi5ELb1ELb1E" in k:
            assert r["Occupancy"] >= 2, (k, r)
    mx = remarks("mx.hip", vg)
    big = [r for k, r in mx.items() if "gemm_mx_kernelILi2ELi4ELi8ELi4" in k or "gemm_mx_pp_kernel" in k]
    # the lockstep 256 x 256 kernel + the ping-pong kernel with the general epilogue, its four fixed kinds (PLAIN, GELU2, AUX, RES) and the two with
    # the fused MX output (GELU2 -> MX, AUX -> MX)
    assert len(big) == 8 and all(b["ScratchSize"] == 0 and b["VGPRs"] + b.get("AGPRs", 0) <= 256 for b in big), big
    # round 6: the decode-row GEMM (gemm.hip skinny2).  Nothing in scratch (the CW = 4 LayerNorm form spilled 84 bytes per lane until the fragments were
    # kept packed across the statistics' barriers), and the cfg2 forms -- 8 waves x 5 chunks -- keep two workgroups per CU (4 waves per SIMD)
    g1 = remarks("gemm.hip", vg)
    sk = {k: r for k, r in g1.items() if "gemm_skinny2_kernel" in k or "gemm_skinny2_long_kernel" in k or "gemm_skinny2_ln_kernel" in k}
    # 7 plain (8 waves x 1 ... 5 chunks, 16 x 3, 4) + the long-K form + the fused-LayerNorm forms: 5 K of 8 waves x units-per-wave 1, 2, 4, 7, 10 and
    # 2 K of 16 waves x 1, 2, 4, 7, + the two-tile forms of K = 2560 (5) and K = 4096 (4)
    assert len(sk) == 7 + 1 + 5 * 5 + 2 * 4 + 5 + 4, sorted(sk)
    import re
    for k, r in sk.items():
        assert r["ScratchSize"] == 0, (k, r)
        m = re.search(r"gemm_skinny2_ln_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)EE", k)
        if "ILi8E" in k and not (m and int(m.group(3)) == 10):        # two workgroups per CU (4 waves per SIMD); ten units per wave (M >= 12 at K = 2560) take one
            assert r["Occupancy"] >= 4, (k, r)


def test_decode_row_gemm_split_and_decode_attention_slots_host_rules():
    """host arithmetic of the decode path, callable without a GPU: (1) unimp_gemm_skinny_rows -- the weight rows per MFMA tile minimise the busiest CU's
    rows below 512 sixteen-row tiles (10 x 256 workgroups for N = 2560, 15 x 512 for 7680, 4 x 128 for 512), stay 16 from there on with one tile per
    workgroup, and with two tiles allowed (the fused-LayerNorm forms at M >= 4) N = 10 240 becomes 2 x 10 rows x 512 workgroups, 7680 2 x 15 x 256;
    the head keeps 16-row single tiles; (2) unimp_attn_decode_step_slots -- one partial slot per 128-key chunk, plus one per 32-key prefix chunk with
    beam groups while the merge can hold them (64), none beyond."""
    import ctypes as C
    from unimp_amd import _lib
    L = _lib.lib()
    def rows(N, mt):
        nt = C.c_int(0)
        return L.unimp_gemm_skinny_rows(N, mt, C.byref(nt)), nt.value
    assert rows(2560, 1) == (10, 1) and rows(7680, 1) == (15, 1) and rows(512, 1) == (4, 1) and rows(10240, 1) == (16, 1) and rows(74053, 1) == (16, 1)
    assert rows(10240, 2) == (10, 2) and rows(7680, 2) == (15, 2) and rows(2560, 2) == (10, 1) and rows(74053, 2) == (16, 1)
    for N in (24, 1005, 2560, 5120, 7680, 10240, 12288):
        for mt in (1, 2):
            R, nt = rows(N, mt)
            assert 4 <= R <= 16 and nt in (1, 2) and nt <= mt
            wgs = -(-N // (R * nt))
            assert -(-wgs // 256) * R * nt <= -(-(-(-N // 16)) // 256) * 16, (N, mt, R, nt)       # never more rows on the busiest CU than 16-row tiles put there
    assert L.unimp_attn_decode_step_slots(10, 32, 527, 1) == 5 and L.unimp_attn_decode_step_slots(10, 32, 527, 10) == 5 + 17
    assert L.unimp_attn_decode_step_slots(10, 32, 4000, 10) == 32 and L.unimp_attn_decode_splits(10, 32, 4000) == 32


def test_gemm_kernel_code_fits_the_instruction_cache():
    """The large-tile GEMM kernels must stay inside the 64 KiB instruction cache two CUs share: with the row-group loop of
    the epilogue unrolled around the fully general per-element code they were 191 KiB, and a 256 x 256 tile spent 15 us in its
    epilogue.  Also: the persistent kernel keeps two waves per SIMD and only its epilogue may touch scratch."""
    import os
    import re
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    bundler, readelf = "/opt/rocm/lib/llvm/bin/clang-offload-bundler", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(hipcc) and os.path.exists(bundler) and os.path.exists(readelf)):
        import pytest
        pytest.skip("no hipcc / llvm tools")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "unimp_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        for name in ("gemm3", "gemm5", "gemm6", "gemm7"):
            obj, hsaco = os.path.join(tmp, name + ".o"), os.path.join(tmp, name + ".hsaco")
            vgpr_form = [] if name == "gemm7" else ["-mllvm", "-amdgpu-mfma-vgpr-form"]       # gemm7 keeps its accumulators in AGPRs (Makefile)
            r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", *vgpr_form, f"-I{root}/include",
                                f"-I{src}", "-c", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", obj,
                                os.path.join(src, name + ".hip")], capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stderr[-2000:]
            subprocess.run([bundler, "--unbundle", "--type=o", f"--input={obj}", "--targets=hip-amdgcn-amd-amdhsa--gfx950",
                            f"--output={hsaco}"], check=True, capture_output=True)
            syms = subprocess.run([readelf, "-s", hsaco], capture_output=True, text=True, check=True).stdout
            sizes = {m.group(2): int(m.group(1)) for m in re.finditer(r"^\s*\d+:\s+[0-9a-f]+\s+(\d+)\s+FUNC\s+\S+\s+\S+\s+\S+\s+(\S+)", syms, re.M)}
            kern = {k: v for k, v in sizes.items() if "bf16_kernel" in k}
            assert kern, syms[:500]
            # gemm6 instantiates every epilogue kind for each of its four 32-row sub-passes; a tile executes one kind only
            limit = (112 if name == "gemm6" else 64) * 1024
            for k, v in kern.items():
                if name == "gemm7" and k.endswith("ELin1EEv11Gemm2Params"):
                    # gemm7's run-time-dispatch kernels (EPI = -1: accumulate / f32 / generic forms, i.e. the weight gradients with K = all tokens):
                    # three copies of the hand-ordered loop body (steady state, second-to-last, last stage) + every epilogue kind = 94 KiB; the steady
                    # loop is 3.5 KiB of it and the epilogue runs once per 512-stage tile
                    assert v <= 100 * 1024, (k, v)
                    continue
                assert v <= limit, (k, v)
            if name == "gemm7":
                scr = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
                agpr = [int(x) for x in re.findall(r"AGPRs: (\d+)", r.stderr)]
                assert scr and max(scr) == 0 and agpr and min(agpr) == 256, (scr, agpr)         # no spill; the 64 accumulator tiles live in AGPRs
            if name == "gemm6":
                occ = [int(x) for x in re.findall(r"Occupancy \[waves/SIMD\]: (\d+)", r.stderr)]
                scr = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
                assert occ and min(occ) >= 2 and max(scr) <= 512, (occ, scr)


def test_attention3_schedule_is_the_generators_output():
    """unimp_amd/csrc/attention3_sched.inc is generated (and its hazard / wait / early-clobber rules checked) by tools/gen_attn3.py: the
    committed file must be exactly what the script produces -- a hand edit would bypass every check the schedule relies on."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_attn3.py"), "--check"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
