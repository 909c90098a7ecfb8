"""One-GPU stand-in for the question the 8-GPU run will answer: do RCCL's kernels get CUs while backward's GEMMs
run, and what do they cost each other?  A streaming kernel with RCCL's footprint (G persistent 512-thread workgroups, no LDS,
read + read + write) runs on a second stream next to the step's dominant GEMM kernels (ping-pong pp256: one 8-wave workgroup per
CU with 128 KiB LDS; persistent ping-pong pp256p: one resident workgroup per CU for the kernel's lifetime; 8-wave w8).
Reports: GEMM alone, stream alone, both together (each one's slowdown), and the same with the GEMM grid capped at CUs - k
(UNIMP_GEMM_RESERVE_CUS) if the library was built with that knob."""
import ctypes as C
import os
import subprocess
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unimp_amd import ops        # noqa: E402

so = os.path.join(ROOT, "tools", "micro", "libinterfere.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(so.replace("libinterfere.so", "interfere.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "tools", "micro", "interfere.hip")])
lib = C.CDLL(so)
lib.interfere_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]

dev = torch.device("cuda")
bf = torch.bfloat16
side = torch.cuda.Stream()
NB = 256 << 20                                   # one default bucket
src = torch.zeros(NB // 4, dtype=torch.int32, device=dev)
dst = torch.zeros(NB // 4, dtype=torch.int32, device=dev)


def timed(fn_main, fn_side, reps_main):
    """events on both streams; returns (ms main per call, ms side total)"""
    torch.cuda.synchronize()
    m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if fn_side is not None:
        with torch.cuda.stream(side):
            s0.record()
            fn_side()
            s1.record()
    m0.record()
    if fn_main is not None:
        for _ in range(reps_main):
            fn_main()
    m1.record()
    torch.cuda.synchronize()
    return (m0.elapsed_time(m1) / max(1, reps_main) if fn_main is not None else 0.0,
            s0.elapsed_time(s1) if fn_side is not None else 0.0)


def main():
    shapes = [("LM up-proj fwd (KC,KC)", 32768, 10240, 2560, False, False),
              ("LM dX through up-proj (KC,KS)", 32768, 2560, 10240, False, True),
              ("gated dW (KS,KS)", 10240, 2560, 32768, True, True)]
    print(f"{'shape':34s} {'variant':7s} {'G':>3s} | gemm alone ms  with stream ms (x) | stream alone GB/s  with gemm GB/s (x)")
    for name, M, N, K, aks, bks in shapes:
        a = torch.randn((K, M) if aks else (M, K), device=dev).to(bf)
        b = torch.randn((K, N) if bks else (N, K), device=dev).to(bf)
        out = torch.empty((M, N), dtype=bf, device=dev)
        for variant in ("pp256", "pp256p", "w8"):
            if variant == "pp256p" and (aks or bks) and aks != bks:
                pass
            g = lambda: ops.gemm(a, b, a_ks=aks, b_ks=bks, out=out, variant=variant)
            try:
                g()
            except Exception as e:       # noqa: BLE001
                print(f"{name:34s} {variant:7s} unsupported: {e}")
                continue
            reps = 12
            t_alone, _ = timed(g, None, reps)
            for G in (16, 32, 64):
                # size the side kernel to run about as long as the GEMM loop: 3 * NB bytes per rep
                _, s_probe = timed(None, lambda: lib.interfere_launch(dst.data_ptr(), src.data_ptr(), NB, G, 1, side.cuda_stream), 0)
                sreps = max(1, int(t_alone * reps / max(s_probe, 1e-3)))
                sf = lambda: lib.interfere_launch(dst.data_ptr(), src.data_ptr(), NB, G, sreps, side.cuda_stream)
                _, s_alone = timed(None, sf, 0)
                t_both, s_both = timed(g, sf, reps)
                bw = lambda ms: 3 * NB * sreps / ms / 1e6
                print(f"{name:34s} {variant:7s} {G:3d} | {t_alone:8.3f}  {t_both:8.3f} ({t_both / t_alone:4.2f}x) | "
                      f"{bw(s_alone):8.0f}  {bw(s_both):8.0f} ({s_both / s_alone:4.2f}x)", flush=True)


if __name__ == "__main__":
    main()
