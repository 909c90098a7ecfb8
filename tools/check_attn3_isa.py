"""Audit of the compiled attention3.hip loops (hipcc -save-temps .s): inside every loop that carries the generated schedule the
compiler may add scalar bookkeeping only -- a vector move (a live-range split) could read an MFMA result the compiler does not
know is still in flight; scratch traffic or a scalar load would break the counted waits.  Prints one line per loop; exit 1 on a finding."""
import re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "build/obj/attention3-hip-amdgcn-amd-amdhsa-gfx950.s"
lines = open(path).read().split("\n")
blocks, cur = [], None
for ln in lines:
    if re.match(r"^\.LBB\d+_\d+:", ln):
        cur = [ln.split(":")[0], []]
        blocks.append(cur)
    elif cur is not None:
        cur[1].append(ln)
bad = 0
for name, body in blocks:
    n_mfma = sum("v_mfma" in l for l in body)
    if n_mfma < 20:
        continue
    in_asm, movs, other = False, [], []
    for l in body:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True; continue
        if t.startswith(";;#ASMEND"):
            in_asm = False; continue
        if in_asm or not t or t.startswith(";"):
            continue
        op = t.split()[0]
        if op.startswith("v_mov") or op.startswith("v_accvgpr") or op.startswith("v_pk_mov"):
            movs.append(t)
        if op.startswith("scratch_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
            other.append(t)
    nops = sum(l.strip().startswith("s_nop") for l in body)
    print(f"{name}: {n_mfma} MFMAs, {len(movs)} compiler vector moves, {len(other)} scratch / scalar loads, {nops} s_nop")
    for m in movs[:6]:
        print("    ", m)
    # the one legitimate move: a scalar broadcast (v_mov_b32 vN, sM) for the DMA address arithmetic
    real = [m for m in movs if not re.match(r"v_mov_b32_e32 v\d+, s\d+", m)]
    bad += len(real) + len(other)
sys.exit(1 if bad else 0)
