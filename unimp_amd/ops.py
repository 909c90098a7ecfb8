"""Raw (non-autograd) wrappers: torch tensors in, HIP kernels through the C ABI, torch tensors out.

Every tensor must live on a HIP device; nothing here has a CPU or eager-PyTorch fallback.
"""
import ctypes as C
import json
import os
import torch

from . import _lib
from ._lib import GemmDesc, AttnDesc, MxGemmDesc, check

_os_env = os.environ.get

ACT = {None: 0, "none": 0, "gelu": 1, "quick_gelu": 2, "relu": 3, "silu": 4, "deriv": 5, "deriv_u8": 6}
MASK_NONE, MASK_CAUSAL, MASK_SEGMENT = 0, 1, 2
bf16 = torch.bfloat16


GEMM_PROFILE = None      # bench.py sets this to a list: (start_event, end_event, flops) per GEMM launch
GEMM_AUTOTUNE = True     # time the kernel variants once per (M, N, K, layout) on scratch operands and keep the fastest
GEMM_VARIANTS = {"auto": 0, "v1": 1, "dma256": 2, "dma128": 3, "pp256": 4, "pp128": 5, "skinny": 6, "w4": 7, "w8": 8, "pp256p": 9,
                 "pp256x": 10, "pp128x": 11, "pp256px": 12, "pp256a": 13, "pp128a": 14, "w4x": 15, "w4x_s1": 16, "w4x_pf": 17, "pp256b": 18, "pp128b": 19, "dw": 20}
_GEMM_CHOICE = {}
# The autotune table is DATA: the one the published numbers were measured with ships INSIDE the package (unimp_amd/gemm_autotune_gfx950.json;
# profiles/gemm_autotune_gfx950.json is the measured copy the profiles cite -- tests/test_cabi_cpu.py keeps the two identical),
# keyed by (M, N, K, a k-strided, b k-strided, epilogue reads an [M, N] input), and is loaded by default, so the variant per shape --
# hence the fp32 summation order, hence the bits, and +-2 % of throughput -- does not depend on timing noise of the box; only a
# shape that is not in the table is tuned live.  UNIMP_GEMM_TUNE_FILE names another table; UNIMP_GEMM_TUNE_WRITE=1 writes newly
# tuned entries back to it (rank 0 only, tmp file + rename); UNIMP_GEMM_TUNE_FILE="" starts from an empty table.
_TUNE_DEFAULT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_autotune_gfx950.json")
_TUNE_FILE = os.environ.get("UNIMP_GEMM_TUNE_FILE", _TUNE_DEFAULT)
_TUNE_WRITE = os.environ.get("UNIMP_GEMM_TUNE_WRITE", "0") == "1"
TUNE_MISSES = []         # keys tuned live in this process (bench.py reports the count)


def _load_tune_table(path):
    if not path or not os.path.exists(path):
        return {}
    try:
        with open(path) as f:
            raw = json.load(f)
        return {tuple(json.loads(k)): int(v) for k, v in raw.items()}      # key[4] is bool (b k-strided) or "pk" (packed-B decision)
    except (ValueError, OSError, TypeError) as e:          # a truncated / corrupt table is not fatal: tune live
        import warnings
        warnings.warn(f"ignoring unreadable GEMM autotune table {path!r}: {e}")
        return {}


def _save_tune_table(path):
    if int(os.environ.get("RANK", "0")) != 0:
        return
    tmp = f"{path}.tmp.{os.getpid()}"
    with open(tmp, "w") as f:
        json.dump({json.dumps([int(k[0]), int(k[1]), int(k[2]), bool(k[3]), k[4] if k[4] == "pk" else bool(k[4]),
                               k[5] if k[5] == "out2" else bool(k[5])]): v
                   for k, v in sorted(_GEMM_CHOICE.items(), key=lambda kv: str(kv[0])) if len(k) == 6}, f, indent=0)
    os.replace(tmp, path)


_GEMM_CHOICE.update(_load_tune_table(_TUNE_FILE))


def _launch_gemm(d, variant):
    check(_lib.lib().unimp_gemm_bf16_variant(C.byref(d), variant, _stream()), "gemm")


def _tune_gemm(M, N, K, a_ks, b_ks, device, reads_mn=False):
    """pick the fastest variant for this problem class, on scratch data.  Three epilogue classes per shape: plain (False), with a
    per-element [M, N] input (True: residual / stored act'(z)) and with a second [M, N] output ("out2": the up-projection's
    GELU + stored GELU') -- the persistent kernel hides a tile's prologue under its epilogue and wins or loses by 10-20 %
    depending on which epilogue that is.  Runs once per key, outside graph capture."""
    key = (M, N, K, a_ks, b_ks, "out2" if reads_mn == "out2" else bool(reads_mn))
    v = _GEMM_CHOICE.get(key)
    if v is not None:
        return v
    if not GEMM_AUTOTUNE or torch.cuda.is_current_stream_capturing():
        return 0
    if 64 < M < 512 and N >= 128 and K >= 256 and K % 64 == 0 and not a_ks:
        # a prompt's rows (the prefill of a generate() call: 469 tokens at the reference's sizes; any M, so no table and no live tuning): the 128 x 128
        # whole-row-A ping-pong kernel -- measured fastest of ten variants at M = 128 / 256 / 469 on all four decoder shapes (v1 + 15 ... 30 %:
        # [469, 10240] x K = 2560 59 -> 43 us, [469, 2560] x K = 10240 169 -> 131 us; tools/bench_gemm_prefill_rows.py).  Same bits as v1.
        _GEMM_CHOICE[key] = 14
        return 14
    cands = [1] if (M < 512 or N < 128 or K < 128) else ([1, 4, 5, 8, 9, 10, 11, 12] if reads_mn == "out2" else [1, 4, 5, 2, 3, 8, 9, 10, 11, 12])   # "out2": the uint8 derivative
    if len(cands) > 1 and not a_ks and K % 64 == 0 and K >= 256:
        cands += [13, 14, 18, 19]               # whole-row A staging (k-contiguous A, 64-k stages); other forms would only re-run the X kernels.
        #                                         18 / 19: the same with the L phase spelled in asm + the L2 prefetch of the panels' shares
    if len(cands) > 1 and (not a_ks or b_ks) and K % 64 == 0 and K >= 128 and M >= 1024 and N >= 256:
        cands += [15] if a_ks else [15, 17]     # w4x (gemm7.hip): one wave per SIMD, hand-ordered two-set loop; 17: with the L2 prefetch.  Same bits as the others
    if len(cands) == 1:
        _GEMM_CHOICE[key] = cands[0]
        return cands[0]
    r8 = lambda n: (n + 7) // 8 * 8
    a = torch.randn((K, r8(M)) if a_ks else (M, r8(K)), device=device, dtype=torch.float32).to(bf16)
    b = torch.randn((K, r8(N)) if b_ks else (N, r8(K)), device=device, dtype=torch.float32).to(bf16)
    ldc = r8(N)
    c = torch.empty((M, ldc), dtype=bf16, device=device)
    d = GemmDesc()
    d.A, d.B, d.C, d.M, d.N, d.K = a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K
    d.lda, d.ldb, d.ldc, d.a_kstrided, d.b_kstrided, d.alpha = a.stride(0), b.stride(0), ldc, int(a_ks), int(b_ks), 1.0
    if reads_mn == "out2":                     # the form the MLP blocks use: act(z) and act'(z) as uint8
        r = torch.empty((M, ldc), dtype=torch.uint8, device=device)
        bias = torch.zeros(r8(N), dtype=bf16, device=device)
        d.pre, d.ldpre, d.pre_deriv, d.act, d.bias = r.data_ptr(), ldc, 2, ACT["gelu"], bias.data_ptr()
    elif reads_mn:
        r = torch.randn((M, ldc), device=device, dtype=torch.float32).to(bf16)
        d.res, d.ldres = r.data_ptr(), ldc
    # two interleaved rounds over the candidates (a candidate's two bursts are not back to back: clock and cache state drift during a
    # tuning pass), the better burst of each counts.  Round 5: with a dozen candidates within a few per cent of each other one burst of
    # three launches picked by noise as often as by speed.
    ok = []
    for v in cands:
        try:
            _launch_gemm(d, v)                   # warm: code, attribute calls
            ok.append(v)
        except _lib.UnimpHipError:               # a candidate the library refuses for this problem (operand size limits, ...) is not a candidate
            pass
    cands = ok or [1]
    times = {v: float("inf") for v in cands}
    for _ in range(2):
        for v in cands:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                _launch_gemm(d, v)
            e1.record()
            e1.synchronize()
            times[v] = min(times[v], e0.elapsed_time(e1))
    best = min(cands, key=lambda v: times[v])
    _GEMM_CHOICE[key] = best
    _note_live_tune(key)
    if _TUNE_FILE and _TUNE_WRITE:
        _save_tune_table(_TUNE_FILE)
    return best


def _note_live_tune(key):
    """a shape the shipped table lacks was timed on this box: said ONCE per process (the pick -- hence the fp32 summation order of that
    shape and a few per cent of its speed -- then depends on this run's timing; UNIMP_GEMM_TUNE_WRITE=1 keeps the picks)"""
    if not TUNE_MISSES and not _TUNE_WRITE:
        import warnings
        warnings.warn(f"unimp_amd: GEMM shape {key} is not in the autotune table {_TUNE_FILE!r}: tuning live (this and any further such "
                      "shape; set UNIMP_GEMM_TUNE_WRITE=1 to record the picks, UNIMP_GEMM_TUNE_FILE to name another table)", stacklevel=3)
    TUNE_MISSES.append(key)


# Data parallelism with more than one rank (set by train.Trainer): RCCL's all-reduce kernels take a handful of CUs for the whole backward
# pass.  A persistent GEMM (one workgroup per CU walking its tiles: variants pp256p / pp256px) then has CUs whose workgroup shares its CU
# and finishes last -- measured 1.55-1.64x on the kernel beside a 16-workgroup streaming kernel, against 1.1-1.4x for the one-tile-per-
# workgroup kernels, whose tiles the dispatcher simply hands to whichever CU is free (profiles/r03_interference_rccl_footprint.txt).  The
# autotuned choice is mapped to the non-persistent twin with the same main loop; every variant produces the same bits.
AVOID_PERSISTENT = False
_PERSISTENT_TWIN = {9: 4, 12: 10}
_avoid_persistent_holders = 0


def avoid_persistent_acquire():
    """a data-parallel Trainer with more than one rank comes up: AVOID_PERSISTENT holds while at least one such trainer lives"""
    global AVOID_PERSISTENT, _avoid_persistent_holders
    _avoid_persistent_holders += 1
    AVOID_PERSISTENT = True


def avoid_persistent_release():
    """... and lets go (Trainer.close / dp.remove): the flag clears with the LAST holder, whatever the order trainers are closed in"""
    global AVOID_PERSISTENT, _avoid_persistent_holders
    _avoid_persistent_holders = max(0, _avoid_persistent_holders - 1)
    if _avoid_persistent_holders == 0:
        AVOID_PERSISTENT = False

_IN_BACKWARD = 0


def backward_scope(fn):
    """decorator for ``Function.backward`` bodies: the GEMMs they issue are backward GEMMs (dX / dW) and may run under split-K"""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        global _IN_BACKWARD
        _IN_BACKWARD += 1
        try:
            return fn(*a, **k)
        finally:
            _IN_BACKWARD -= 1
    return wrapped


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """the current HIP stream's handle.  torch.cuda.current_stream() builds a Stream object through three Python layers -- 8 us per call, once per
    launch: 5 of the 29 ms of host time of a 469-token prefill; the raw getter returns the same handle in a fraction of a microsecond."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _dev(t):
    if not t.is_cuda:
        raise _lib.UnimpHipError("unimp_amd ops need tensors on a HIP device (no CPU fallback exists)")
    return t


def _p(t):
    return 0 if t is None else _dev(t).data_ptr()


def _mat(t):
    """2-D bf16 view with unit inner stride; returns (tensor, ld)."""
    assert t.dim() == 2 and t.dtype == bf16, (t.shape, t.dtype)
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t, (t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1]))


class PackedB:
    """pre-packed B operand of a frozen weight (include/unimp_hip.h: unimp_pack_b_bf16): fragment-ordered image + logical (N, K)."""
    __slots__ = ("buf", "N", "K")

    def __init__(self, buf, N, K):
        self.buf, self.N, self.K = buf, N, K


def pack_b(b, b_ks=False):
    """b: [N, K] (or [K, N] if b_ks) bf16 -> PackedB."""
    b, ldb = _mat(b)
    (K, N) = b.shape if b_ks else b.shape[::-1]
    nbytes = _lib.lib().unimp_pack_b_bytes(N, K)
    buf = torch.empty(nbytes // 2, dtype=bf16, device=b.device)
    check(_lib.lib().unimp_pack_b_bf16(b.data_ptr(), ldb, N, K, int(b_ks), buf.data_ptr(), _stream()), "pack_b")
    return PackedB(buf, N, K)


PACKED_MIN_M = 1024       # a packed B operand is only tried on problems the ping-pong kernels serve


def _tune_packed(M, N, K, a_ks, device, reads_mn, unpacked_variant, b_ks):
    """packed-B ping-pong (pp256 / pp128) against the table's choice for the SAME problem with B through the LDS; returns the
    packed variant to use or 0 (keep the unpacked kernel).  Key: (M, N, K, a_ks, "pk", reads_mn)."""
    key = (M, N, K, a_ks, "pk", bool(reads_mn))
    v = _GEMM_CHOICE.get(key)
    if v is not None:
        return v
    if not GEMM_AUTOTUNE or torch.cuda.is_current_stream_capturing():
        return 4 if N >= 256 else 5
    r8 = lambda n: (n + 7) // 8 * 8
    a = torch.randn((K, r8(M)) if a_ks else (M, r8(K)), device=device, dtype=torch.float32).to(bf16)
    b = torch.randn((K, r8(N)) if b_ks else (N, r8(K)), device=device, dtype=torch.float32).to(bf16)
    pk = pack_b(b[:, :N] if b_ks else b[:, :K], b_ks)
    ldc = r8(N)
    c = torch.empty((M, ldc), dtype=bf16, device=device)
    d = GemmDesc()
    d.A, d.C, d.M, d.N, d.K = a.data_ptr(), c.data_ptr(), M, N, K
    d.lda, d.ldc, d.a_kstrided, d.alpha = a.stride(0), ldc, int(a_ks), 1.0
    if reads_mn:
        r = torch.randn((M, ldc), device=device, dtype=torch.float32).to(bf16)
        d.res, d.ldres = r.data_ptr(), ldc

    def run(v, packed):
        if packed:
            d.B, d.ldb, d.b_kstrided = pk.buf.data_ptr(), 0, 2
        else:
            d.B, d.ldb, d.b_kstrided = b.data_ptr(), b.stride(0), int(b_ks)
        _launch_gemm(d, v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            _launch_gemm(d, v)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1)
    best, best_t = 0, run(unpacked_variant, False)
    for v in ([4, 5] if N >= 256 else [5]):
        t = run(v, True)
        if t < best_t * 0.99:
            best, best_t = v, t
    _GEMM_CHOICE[key] = best
    _note_live_tune(key)
    if _TUNE_FILE and _TUNE_WRITE:
        _save_tune_table(_TUNE_FILE)
    return best


TAIL_SPLIT = os.environ.get("UNIMP_GEMM_TAIL_SPLIT", "1") != "0"


ROPE_VARIANTS = (4, 9, 10, 12, 13, 15, 18)  # pp256 / pp256p, their one-set forms and the whole-row-A form: the kernels with a rotary-epilogue instantiation (gemm3.hip, gemm6.hip)


ROPE_MIN_M = 256                # rows (one tile row) from which the QKV projection takes the rotary epilogue: low enough that a training
                                # sample's bits do not depend on its batch size (b = 1, L = 512 and b = 64 both take it)


def gemm_rope_variant(M, N, K, b_ks, device):
    """variant that serves the rotary epilogue for an [M, N, K] QKV projection, or None.  Whether the rotation is FUSED is decided
    on static criteria only (shape and layout) -- never on the autotune table or on timing: the fused epilogue computes cos / sin
    on the fly (v_exp / v_sin / v_cos) where the separate pass reads the fp32 tables, so a table- or timing-dependent choice would
    make a sample's bits depend on the box.  The table only picks WHICH of the two kernels with the same rotary
    arithmetic runs (ping-pong or persistent ping-pong)."""
    if M < ROPE_MIN_M or N < 128 or N % 8 or M >= 1 << 24:
        return None
    v = _tune_gemm(M, N, K, False, bool(b_ks), device, False)          # b_ks: the transposed copy of a frozen projection (same bits)
    return v if v in ROPE_VARIANTS else 4


def _splitk_count(M, N, K, tiles256):
    """slices for a split-K GEMM on 256 x 256 tiles.  More slices fill the 256 CUs better and shorten each block, but every
    slice writes and re-reads an fp32 [M, N] slab: a block costs ~6 us + 0.8 us per 32 k, the slabs move at ~4 TB/s (measured:
    [2560, 512, 32768] with 25 slices 129 us, of which 65 us slab traffic)."""
    best, best_t = 2, float("inf")
    for S in range(2, 33):
        ks = (-(-K // S) + 63) // 64 * 64
        n = -(-K // ks)
        if n != S or ks < 512:
            continue
        rounds = -(-tiles256 * S // 256)
        t = rounds * (6.0 + 0.8 * ks / 32) + S * M * N * 4 * 2 / 4e6 + 5.0
        if t < best_t:
            best, best_t = S, t
    return best


def _tail_split_plan(M, N, K):
    """Weight-gradient GEMM whose 256 x 256 tiles fill the 256 CUs once and then leave a mostly empty second round (400 tiles:
    the second round is 44 % idle): cut the output along the dimension with the finer tile granularity into a part of <= 256
    tiles (one full round, ordinary kernel) and a remainder that runs under split-K, its slices sized to fill whole rounds.
    Returns (axis, cut_elements, splits) or None.  Model: a tile costs T = its FLOPs at the deep-K main-loop rate, a split-K
    slice T / S, the slabs one write + one read at 4 TB/s."""
    nbm, nbn = (M + 255) // 256, (N + 255) // 256
    tiles = nbm * nbn
    if not (256 < tiles < 512) or K < 8192:
        return None
    T = 2.0 * 256 * 256 * K / 5.4e12
    base = 2.0 * T
    best = None
    for axis, step, n_line in (("m", nbn, nbm), ("n", nbm, nbn)):
        k = min(256 // step, n_line - 1)
        if k < 1:
            continue
        ta, tb = k * step, tiles - k * step
        for S in range(2, 9):
            if K // S < 1024:
                break
            rounds = -(-tb * S // 256)
            cost = T + rounds * T / S + tb * S * 262144 * 2 / 4e12
            if best is None or cost < best[0]:
                best = (cost, axis, k * 256, S)
    if best is None or best[0] > 0.92 * base:
        return None
    return best[1], best[2], best[3]


def gemm(a, b, *, a_ks=False, b_ks=False, bias=None, act=None, res=None, aux=None, dact=None, pre=None,
         gate=None, alpha=1.0, out=None, out_f32=False, accumulate=False, ldc=None, variant=None, pre_deriv=False, b_pk=None,
         _splits=None, rope=None, ln=None):
    """C[M,N] = epi(alpha * A B^T).  a: [M,K] (or [K,M] if a_ks); b: [N,K] (or [K,N] if b_ks).
    b_pk: optional PackedB image of the same b (frozen weights): used when the packed ping-pong kernel measured faster.
    ln = (gamma, beta or None, eps): decode rows only (skinny_ln_ok(M, K)) -- the rows of a are layer-normalised inside the kernel."""
    a, lda = _mat(a)
    b, ldb = _mat(b)
    if a_ks:
        K, M = a.shape
    else:
        M, K = a.shape
    if b_ks:
        Kb, N = b.shape
    else:
        N, Kb = b.shape
    assert K == Kb, (a.shape, b.shape, a_ks, b_ks)
    if out is None:
        ldc = ldc or N
        out = torch.empty((M, ldc), dtype=torch.float32 if out_f32 else bf16, device=a.device)
        if ldc != N:
            out = out[:, :N]
    if (TAIL_SPLIT and variant is None and _splits is None and a_ks and b_ks and bias is None and res is None and aux is None
            and pre is None and act is None and dact is None and N % 4 == 0 and out.stride(0) % 4 == 0):
        plan = _tail_split_plan(M, N, K)
        if plan is not None:
            axis, cut, S = plan
            kw = dict(a_ks=True, b_ks=True, gate=gate, alpha=alpha, accumulate=accumulate)
            if axis == "m":
                gemm(a[:, :cut], b, out=out[:cut], _splits=0, **kw)
                gemm(a[:, cut:], b, out=out[cut:], _splits=S, **kw)
            else:
                gemm(a, b[:, :cut], out=out[:, :cut], _splits=0, **kw)
                gemm(a, b[:, cut:], out=out[:, cut:], _splits=S, **kw)
            return out
    tuned_rope = False               # the rotary path resolves its variant through the tuner too: the persistent-twin mapping applies to it
    d = GemmDesc()
    d.A, d.B, d.C = a.data_ptr(), b.data_ptr(), _dev(out).data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = lda, ldb, out.stride(0)
    d.a_kstrided, d.b_kstrided = int(a_ks), int(b_ks)
    d.bias = _p(bias)
    if res is not None:
        d.res, d.ldres = res.data_ptr(), res.stride(0)
    if aux is not None:
        d.aux, d.ldaux = aux.data_ptr(), aux.stride(0)
    if pre is not None:
        d.pre, d.ldpre = pre.data_ptr(), pre.stride(0)
    d.gate = _p(gate)
    d.alpha = alpha
    d.act, d.dact = ACT[act], ACT[dact]
    d.out_f32, d.accumulate = int(out.dtype == torch.float32), int(accumulate)
    d.pre_deriv = int(pre_deriv)
    if pre is not None and pre.dtype == torch.uint8:         # act'(z) in 8 bits (include/unimp_hip.h: pre_deriv = 2, ldpre in bytes)
        assert pre_deriv and act is not None, "a uint8 second output holds the activation's derivative"
        d.pre_deriv = 2
    if aux is not None and aux.dtype == torch.uint8:
        assert dact in ("deriv", "deriv_u8"), "a uint8 aux operand is a stored derivative"
        d.dact = ACT["deriv_u8"]
    if rope is not None:             # rotary epilogue (include/unimp_hip.h): dict(rot, hd, period, span, L, log2_base); see gemm_rope_variant()
        d.rope_rot, d.rope_hd, d.rope_period, d.rope_span, d.rope_L = rope["rot"], rope["hd"], rope["period"], rope["span"], rope["L"]
        d.rope_log2_base = rope["log2_base"]
        if rope.get("pos") is not None:       # packed rows: int32 [M] position of every row (instead of row % L)
            assert rope["pos"].dtype == torch.int32 and rope["pos"].numel() >= M and rope["pos"].is_contiguous()
            d.rope_pos = rope["pos"].data_ptr()
        _splits = 0
        if variant is None:
            tuned_rope = True
            variant = gemm_rope_variant(M, N, K, b_ks, a.device)
            if variant is None:
                raise _lib.UnimpHipError("gemm: no rotary-epilogue kernel for this problem (ops.gemm_rope_variant); run rope_ as a separate pass")
    plain = bias is None and res is None and aux is None and pre is None and act is None and dact is None   # alpha / gate / accumulate only
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    tiles256 = ((M + 255) // 256) * ((N + 255) // 256)
    if ln is not None:
        assert not a_ks and not b_ks and variant is None and skinny_ln_ok(M, K), "gemm(ln=...): decode rows only (ops.skinny_ln_ok)"
        d.ln_gamma, d.ln_beta, d.ln_eps = ln[0].data_ptr(), _p(ln[1]), float(ln[2])
    if variant is None and M <= 64 and not a_ks and not b_ks and K % 64 == 0:
        variant = 6                      # decode rows: the weight-streaming kernel, any epilogue
    # dX form (dy [M, K] x W [K, N], k-strided weight) INSIDE a backward pass with few 256 x 256 tiles and a deep K: at the
    # reference's shipped shape (3 x 512 tokens) [1536, 2560] x K = 10240 / 7680 is 60 tiles -- the 128 x 128 kernel ran them at
    # 0.44-0.48 PFLOP/s (r3d per-shape table).  Forward GEMMs never split (their summation order must not depend on the batch size).
    in_bwd = _IN_BACKWARD > 0      # set by functional.py's Function.backward bodies (ops.backward_scope), NOT sniffed from the autograd engine:
    #                                a forward recomputed inside a backward pass (activation checkpointing) must not take split-K paths
    #                                the original forward did not
    dx_split = (not a_ks) and b_ks and M >= 256 and N >= 256 and K >= 5120 and tiles256 <= 128 and plain and variant is None and in_bwd
    wide = M >= 256 and N >= 256 and a_ks and b_ks   # weight-gradient form only (a forward GEMM of a small batch must not change
    #                                                  its summation order with the batch size); 256 x 256 ping-pong tiles under split-K
    # (K >= 16384 with up to 160 tiles: the LM head's dX over the labeled rows only, [~640, 74 053] x [74 053, 2560] -- 100 tiles, 170 TFLOP/s unsplit)
    # Never in a FORWARD pass (outside autograd's backward, k-contiguous or mixed operands): the slice count follows the tile count, i.e. the
    # batch size, and a sample's bits must not (round 3: the gated cross-attention's to_q, [B L, 512] x K = 2560, ran under 5 slices up to
    # b = 6, 3 at b = 12, none beyond)
    may_split = (a_ks and b_ks) or in_bwd or bool(_splits)           # _splits = -1: allowed here, slice count chosen as in a backward pass
    if (variant is None and plain and _splits != 0 and may_split
            and ((tiles256 <= 128 if wide else (dx_split or tiles <= 96 or (tiles <= 160 and K >= 16384))) or (_splits or 0) > 0)
            and K >= 2048 and N % 4 == 0 and out.stride(0) % 4 == 0):
        # weight gradient of a narrow projection: far fewer tiles than CUs, very deep K -> split-K over the chip
        big = wide or dx_split or (K >= 16384 and M >= 256 and N >= 256)        # the C side runs 256 x 256 tiles whenever M, N >= 256
        splits = _splits if (_splits or 0) > 0 else (_splitk_count(M, N, K, tiles256) if big else max(2, min(32, 320 // tiles, K // 512)))
        slabs = torch.empty((splits, M, N), dtype=torch.float32, device=a.device)
        if GEMM_PROFILE is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        check(_lib.lib().unimp_gemm_bf16_splitk(C.byref(d), splits, slabs.data_ptr(), _stream()), "gemm_splitk")
        if GEMM_PROFILE is not None:
            e1.record()
            GEMM_PROFILE.append((e0, e1, 2.0 * M * N * K, (M, N, K, int(a_ks), int(b_ks), -splits)))
        return out
    forced_pk = {"pk256": 4, "pk128": 5, "dwpk": 20}.get(variant) if isinstance(variant, str) else None
    if forced_pk is not None:
        v = forced_pk
    else:
        v = GEMM_VARIANTS[variant] if isinstance(variant, str) else (variant if variant is not None else
                                                                      _tune_gemm(M, N, K, bool(a_ks), bool(b_ks), a.device,
                                                                                 True if (res is not None or aux is not None) else
                                                                                 ("out2" if pre is not None else False)))
    if AVOID_PERSISTENT and (variant is None or tuned_rope) and v in _PERSISTENT_TWIN:
        v = _PERSISTENT_TWIN[v]
    if v in (15, 16, 17) and variant is None and ((32 if a_ks else M) * lda * 2 >= 1 << 32 or (32 if b_ks else N) * ldb * 2 >= 1 << 32):
        # gemm7 (w4x) addresses its operands with 32-bit offsets: below 4 GiB each.  The table key carries no lda, so a tuned w4x entry
        # can be reached through a wide-ld view: the whole-row-A / one-set ping-pong kernel serves it instead (same bits)
        v = 13 if (not a_ks and K % 64 == 0 and K >= 256) else 10
    if (d.pre_deriv == 2 or d.dact == ACT["deriv_u8"]) and variant is None and v in (0, 2, 3, 6, 7):
        # the uint8 derivative lives in the kernels with the specialised epilogue kinds; 0 = the library's own choice, which may be a DMA variant
        v = 1 if (M < 256 or N < 128) else (4 if N >= 256 else 5)
    if b_pk is not None and (forced_pk is not None or (variant is None and M >= PACKED_MIN_M and N >= 128 and K >= 128)):
        assert b_pk.N == N and b_pk.K == K, (b_pk.N, b_pk.K, N, K)
        vp = forced_pk or _tune_packed(M, N, K, bool(a_ks), a.device, res is not None or aux is not None, v, bool(b_ks))
        if vp:
            d.B, d.ldb, d.b_kstrided, v = b_pk.buf.data_ptr(), 0, 2, vp
    if GEMM_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _launch_gemm(d, v)
        e1.record()
        GEMM_PROFILE.append((e0, e1, 2.0 * M * N * K, (M, N, K, int(a_ks), int(d.b_kstrided), v)))
        return out
    _launch_gemm(d, v)
    return out


def skinny_ln_ok(M, K):
    """may a decode GEMM of M rows and contraction depth K take its LayerNorm fused (gemm(ln=...))?"""
    return bool(_lib.lib().unimp_gemm_skinny_ln_ok(int(M), int(K)))


def layernorm_fwd(x, gamma, beta, eps, *, rms=False, out=None, grp=0, grp_stride=0, grp_off=0):
    x, ldx = _mat(x)
    rows, D = x.shape
    if out is None:
        out = torch.empty((rows, D), dtype=bf16, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(_lib.lib().unimp_layernorm_fwd(x.data_ptr(), ldx, _p(gamma), _p(beta), out.data_ptr(), out.stride(0),
                                          mean.data_ptr(), rstd.data_ptr(), rows, D, eps, int(rms), grp, grp_stride,
                                          grp_off, _stream()), "layernorm_fwd")
    return out, mean, rstd


def layernorm_fwd_mx(x, gamma, beta, eps, *, rms=False):
    """LayerNorm whose output leaves as an MxTensor (quantised along D): the bytes of ``mx_quantize(layernorm_fwd(x)[0])`` without the
    bf16 tensor and the quantiser pass.  returns (MxTensor, mean, rstd)."""
    x, ldx = _mat(x)
    rows, D = x.shape
    q = torch.empty((rows, D), dtype=torch.uint8, device=x.device)
    sc = torch.empty((rows, D // 32), dtype=torch.uint8, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(_lib.lib().unimp_layernorm_fwd_mx(_dev(x).data_ptr(), ldx, _p(gamma), _p(beta), q.data_ptr(), q.stride(0), sc.data_ptr(), sc.stride(0),
                                             mean.data_ptr(), rstd.data_ptr(), rows, D, eps, int(rms), _stream()), "layernorm_fwd_mx")
    return MxTensor(q, sc), mean, rstd


_PARTIAL_BLOCKS = 512       # 2 blocks per CU: the wgrad variant holds 244 VGPRs (occupancy 2), 256 blocks left half the SIMDs with one wave


def layernorm_bwd(dy, x, gamma, mean, rstd, *, dres=None, dy2=None, want_wgrad=False, has_beta=True, rms=False, grp=0,
                  grp_stride=0, grp_off=0, dg_out=None, db_out=None, dx_out=None):
    """returns dx, dgamma, dbeta (bf16; None when not wanted).  dy may be the grouped (concat) buffer.
    dg_out / db_out (bf16 [D] views of a gradient buffer, both or -- without beta -- dg_out alone): the weight gradients are
    ADDED there and None is returned for them."""
    x, ldx = _mat(x)
    rows, D = x.shape
    assert dy.stride(-1) == 1
    dx = torch.empty((rows, D), dtype=bf16, device=x.device) if dx_out is None else dx_out      # dx_out may alias dy (a wave holds its row)
    assert dx.stride(-1) == 1 and dx.shape == (rows, D)
    dg = db = part = None
    acc = 0
    if want_wgrad:
        if dg_out is not None and (db_out is not None or not has_beta):
            dg, db, acc = dg_out, db_out, 1
        else:
            dg = torch.empty(D, dtype=bf16, device=x.device)
            db = torch.empty(D, dtype=bf16, device=x.device) if has_beta else None
        part = torch.empty(_PARTIAL_BLOCKS * 2 * D, dtype=torch.float32, device=x.device)
    check(_lib.lib().unimp_layernorm_bwd(dy.data_ptr(), dy.stride(-2), _p(dy2), dy2.stride(0) if dy2 is not None else 0,
                                          x.data_ptr(), ldx, _p(gamma), _p(mean), _p(rstd),
                                          _p(dres), dres.stride(0) if dres is not None else 0, dx.data_ptr(), dx.stride(0),
                                          _p(dg), _p(db), _p(part), _PARTIAL_BLOCKS, rows, D, int(rms), grp, grp_stride,
                                          grp_off, acc, _stream()), "layernorm_bwd")
    if acc:
        return dx, None, None
    return dx, dg, db


def rope_(x2d, L, heads, head_stride, rot, offs, cos, sin, inverse=False, pos=None):
    """in place on x2d [rows, *]; offs: element offsets of the vectors to rotate inside each head slot.  Position of row r: r % L, or
    pos[r] (int32 [rows]: packed rows)."""
    rows = x2d.shape[0]
    o0 = offs[0]
    o1 = offs[1] if len(offs) > 1 else 0
    if pos is not None:
        assert pos.dtype == torch.int32 and pos.numel() >= rows and pos.is_contiguous()
        check(_lib.lib().unimp_rope_halfsplit_pos(_dev(x2d).data_ptr(), x2d.stride(0), head_stride, rows, pos.data_ptr(), heads, rot, len(offs),
                                                   o0, o1, cos.data_ptr(), sin.data_ptr(), int(inverse), _stream()), "rope")
        return x2d
    check(_lib.lib().unimp_rope_halfsplit(_dev(x2d).data_ptr(), x2d.stride(0), head_stride, rows, L, heads, rot, len(offs),
                                           o0, o1, cos.data_ptr(), sin.data_ptr(), int(inverse), _stream()), "rope")
    return x2d


def decode_rope_append(qkv2d, heads, head_stride, hd, offs, rot, cos_rows, sin_rows, kcache, vcache, pos_idx):
    """decode step: rotate q / k of every row in place (row r with ITS table row) and append the rotated k and v to slot pos_idx[r] of
    the caches [rows, capacity, heads, hd] -- one launch instead of rope_ + two index_put_.  offs = (q_off, k_off, v_off) element
    offsets inside a head slot; rot = 0 appends only."""
    rows = qkv2d.shape[0]
    assert kcache.stride(3) == 1 and vcache.stride() == kcache.stride() and pos_idx.dtype == torch.int64 and pos_idx.is_contiguous()
    if rot:
        assert cos_rows.dtype == torch.float32 and cos_rows.is_contiguous() and sin_rows.is_contiguous() and cos_rows.shape == (rows, rot // 2)
    check(_lib.lib().unimp_decode_rope_append(_dev(qkv2d).data_ptr(), qkv2d.stride(0), head_stride, rows, heads, hd, offs[0], offs[1], offs[2], rot,
                                               _p(cos_rows) if rot else 0, _p(sin_rows) if rot else 0, kcache.data_ptr(), vcache.data_ptr(),
                                               kcache.stride(0), kcache.stride(1), kcache.stride(2), pos_idx.data_ptr(), _stream()), "decode_rope_append")


def beam_topk_ok(logits, K, C):
    V = logits.shape[-1]
    return (logits.is_cuda and logits.dim() == 2 and logits.stride(1) == 1 and logits.dtype in (bf16, torch.float32) and logits.shape[0] % K == 0
            and K <= 16 and C <= 32 and 16 * C <= V <= 98304)


def beam_topk(logits, beam_scores, K, C):
    """one beam-search step's candidate selection: top-C (sorted) of log_softmax(logits) + beam_scores over every prompt's K rows viewed as one
    [K * V] vector -> (scores fp32 [prompts, C], flat indices int64 [prompts, C] = row_in_group * V + token).  Two launches
    (csrc/elementwise.hip beam_topk_a / _b) instead of float copy + log_softmax + add + torch.topk."""
    rows, V = logits.shape
    L = _lib.lib()
    scratch = torch.empty(L.unimp_beam_topk_scratch(rows), dtype=torch.uint8, device=logits.device)
    bs = beam_scores.reshape(-1).float().contiguous()
    out_s = torch.empty((rows // K, C), dtype=torch.float32, device=logits.device)
    out_i = torch.empty((rows // K, C), dtype=torch.int64, device=logits.device)
    check(L.unimp_beam_topk(_dev(logits).data_ptr(), int(logits.dtype == torch.float32), logits.stride(0), rows, V, K, C, bs.data_ptr(), scratch.data_ptr(),
                            out_s.data_ptr(), out_i.data_ptr(), _stream()), "beam_topk")
    return out_s, out_i


def kv_reorder_beams(kv, K, src_local, slot0, pos_idx, max_new):
    """beam search: row j of every beam group takes the generated tail of row src_local[group * K + j] (index inside the group), in place, for all
    layers and both of K / V in ONE launch.  kv [layers, 2, rows, capacity, nh, hd] (the decode cache's tensor); slot0 int32 [groups] = first tail slot
    (the prompt's length); pos_idx int64 [rows] = the slot the current step will write (slots below it hold generated tokens)."""
    Lr, two, R, cap, nh, hd = kv.shape
    assert kv.is_contiguous() and two == 2 and R % K == 0 and src_local.dtype == torch.int64 and slot0.dtype == torch.int32 and pos_idx.dtype == torch.int64
    check(_lib.lib().unimp_kv_reorder_beams(_dev(kv).data_ptr(), kv.stride(1), Lr * 2, kv.stride(2), kv.stride(3), nh * hd, K, R // K,
                                             src_local.data_ptr(), slot0.data_ptr(), pos_idx.data_ptr(), int(max_new), _stream()), "kv_reorder_beams")


_DECODE_STEP_WS = {}


def attn_decode_step(qkv2d, heads, head_stride, hd, offs, rot, cos_rows, sin_rows, kcache, vcache, pos_idx, scale, alibi=None, group=1, shared_len=None,
                     group_mode=0):
    """the self-attention of a cached decode step in ONE launch (csrc/decode_attn.hip attn_decode_step_kernel): rotate the new q / k, write the
    rotated k and v to cache slot pos_idx[r], attend keys [0, pos_idx[r]] -- the bits of decode_rope_append + attn_decode.  Same arguments as
    decode_rope_append (+ scale / ALiBi slopes); qkv2d is left unrotated.  group > 1 with shared_len (int32 [rows // group]): beam search as in
    attn_decode -- the rows of a group hold identical K / V below shared_len[g]; group_mode 0: every row reads them from the group's first row (the
    bits of group = 1, the repeated lines served by the caches), 1: once per prompt by extra workgroups of the same launch.
    Returns o [rows, 1, heads, hd]."""
    rows, cap = qkv2d.shape[0], kcache.shape[1]
    assert kcache.stride(3) == 1 and vcache.stride() == kcache.stride() and pos_idx.dtype == torch.int64 and pos_idx.is_contiguous()
    if rot:
        assert cos_rows.dtype == torch.float32 and cos_rows.is_contiguous() and sin_rows.is_contiguous() and cos_rows.shape == (rows, rot // 2)
    if alibi is not None:
        assert alibi.dtype == torch.float32 and alibi.numel() == heads and alibi.is_contiguous()
    L = _lib.lib()
    splits = L.unimp_attn_decode_splits(rows, heads, cap)
    grouped = group > 1 and shared_len is not None and rows % group == 0 and group <= 16
    nslots = L.unimp_attn_decode_step_slots(rows, heads, cap, group if grouped else 1)
    key = (qkv2d.device, rows, heads, nslots, hd)
    if key not in _DECODE_STEP_WS:          # per shape, kept: the arrival counters must be zero at the first launch and are left zero by every launch
        _DECODE_STEP_WS[key] = (torch.empty(rows * heads * nslots * (hd + 2), device=qkv2d.device, dtype=torch.float32),
                                torch.zeros(rows * heads, device=qkv2d.device, dtype=torch.int32))
    ws, arrived = _DECODE_STEP_WS[key]
    o = torch.empty((rows, 1, heads, hd), device=qkv2d.device, dtype=bf16)
    d = _lib.DecodeStepDesc()
    d.qkv, d.row_stride, d.head_stride = _dev(qkv2d).data_ptr(), qkv2d.stride(0), head_stride
    d.q_off, d.k_off, d.v_off = offs
    d.rows, d.heads, d.hd, d.rot = rows, heads, hd, rot
    d.cos_rows, d.sin_rows = (_p(cos_rows), _p(sin_rows)) if rot else (None, None)
    d.kcache, d.vcache = kcache.data_ptr(), vcache.data_ptr()
    d.c_row_stride, d.c_slot_stride, d.c_head_stride, d.capacity = kcache.stride(0), kcache.stride(1), kcache.stride(2), cap
    d.pos_idx, d.scale, d.alibi_slopes = pos_idx.data_ptr(), scale, _p(alibi)
    d.out, d.o_row_stride, d.o_head_stride = o.data_ptr(), o.stride(0), o.stride(2)
    d.workspace, d.arrived = ws.data_ptr(), arrived.data_ptr()
    if grouped:
        assert shared_len.dtype == torch.int32 and shared_len.numel() == rows // group
        d.group, d.shared_len, d.group_mode = group, shared_len.data_ptr(), int(group_mode)
    else:
        d.group, d.shared_len, d.group_mode = 1, None, 0
    check(L.unimp_attn_decode_step(C.byref(d), _stream()), "attn_decode_step")
    return o


def decode_rope_append_ok(hd, rot, head_stride, offs, qkv2d, kcache):
    return (hd % 8 == 0 and rot % 16 == 0 and rot <= hd and head_stride % 8 == 0 and all(o % 8 == 0 for o in offs) and qkv2d.stride(0) % 8 == 0
            and qkv2d.data_ptr() % 16 == 0 and kcache.data_ptr() % 16 == 0 and all(s_ % 8 == 0 for s_ in kcache.stride()[:3]))


def _fill_attn(d, q, k, v, o, lse, B, H, Sq, Sk, D, scale, mask_mode, kv_len, seg, seg_len, qs, ks, vs, os_, alibi=None):
    d.q, d.k, d.v, d.o, d.lse = q, k, v, o, lse
    d.q_bs, d.q_ss, d.q_hs = qs
    d.k_bs, d.k_ss, d.k_hs = ks
    d.v_bs, d.v_ss, d.v_hs = vs
    d.o_bs, d.o_ss, d.o_hs = os_
    d.B, d.H, d.Sq, d.Sk, d.D = B, H, Sq, Sk, D
    d.scale, d.mask_mode = scale, mask_mode
    d.kv_len, d.seg, d.seg_len = _p(kv_len), _p(seg), seg_len
    if alibi is not None:
        assert alibi.dtype == torch.float32 and alibi.numel() == H and alibi.is_contiguous()
    d.alibi_slopes = _p(alibi)


def _view4(t):
    """t: [B, S, H, D] strided view (last dim contiguous) -> (ptr, (bs, ss, hs))."""
    assert t.dim() == 4 and t.stride(3) == 1 and t.dtype == bf16
    return _dev(t).data_ptr(), (t.stride(0), t.stride(1), t.stride(2))


class PackedRows:
    """the sequences of a batch as row ranges of ONE [rows, H, D] buffer (include/unimp_hip.h: q_row_off / k_row_off): sequence b owns
    rows off[b] .. off[b] + len[b] - 1.  off, len: int32 [B] device tensors; S: the longest sequence (or any bound of it: it sizes the
    launch and the per-row statistics); n: rows in use (the rows behind them are padding of the buffer, never touched by the kernels)."""
    __slots__ = ("B", "S", "off", "len", "n")

    def __init__(self, B, S, off, len_, n):
        assert off.dtype == torch.int32 and len_.dtype == torch.int32 and off.numel() == B and len_.numel() == B
        self.B, self.S, self.off, self.len, self.n = B, S, off.contiguous(), len_.contiguous(), n


def _packed_dims(d, q, k, q_rows, k_rows, kv_len):
    """(B, Sq, Sk) of a call and the descriptor's packed-row fields."""
    B, Sq = (q_rows.B, q_rows.S) if q_rows is not None else (q.shape[0], q.shape[1])
    Sk = k_rows.S if k_rows is not None else k.shape[1]
    if q_rows is not None:
        assert q.shape[0] == 1, "packed rows: q is [1, rows, H, D]"
        d.q_row_off, d.q_len = q_rows.off.data_ptr(), q_rows.len.data_ptr()
    if k_rows is not None:
        assert k.shape[0] == 1 and (kv_len is None or kv_len is k_rows.len), "packed rows: k / v are [1, rows, H, D] and kv_len is their row count"
        assert k_rows.B == B
        d.k_row_off = k_rows.off.data_ptr()
    else:
        assert k.shape[0] == B
    return B, Sq, Sk


def attn_fwd(q, k, v, scale, mask_mode=MASK_NONE, kv_len=None, seg=None, seg_len=0, out=None, alibi=None, q_rows=None, k_rows=None):
    """q [B,Sq,H,D], k/v [B,Sk,H,D] strided views; returns o [B,Sq,H,D] (contiguous unless `out`), lse [B,H,Sq].
    q_rows / k_rows (PackedRows): the query side / key side is [1, rows, H, D] with the sequences as row ranges; o then comes back in
    the packed layout too (rows behind the last sequence zero), lse stays [B, H, S] (entries behind a sequence's length unset); a
    segment mask's ``seg`` is packed like q."""
    d = AttnDesc()
    B, Sq, Sk = _packed_dims(d, q, k, q_rows, k_rows, kv_len)
    if k_rows is not None:
        kv_len = k_rows.len
    H, D = q.shape[2], q.shape[3]
    if out is None:
        out = torch.empty((q.shape[0], q.shape[1], H, D), dtype=bf16, device=q.device)
        if q_rows is not None and q_rows.n < q.shape[1]:
            out[:, q_rows.n:].zero_()
    lse = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device)
    (qp, qs), (kp, ks), (vp, vs), (op, os_) = _view4(q), _view4(k), _view4(v), _view4(out)
    _fill_attn(d, qp, kp, vp, op, lse.data_ptr(), B, H, Sq, Sk, D, scale, mask_mode, kv_len, seg, seg_len, qs, ks, vs, os_, alibi)
    check(_lib.lib().unimp_attn_fwd(C.byref(d), _stream()), "attn_fwd")
    return out, lse


DECODE_ATTN = _os_env("UNIMP_DECODE_ATTN", "1") != "0"       # split-key decode kernel for one-query-row calls (0: the training kernel)
# beam search: read a prompt's K / V once per prompt instead of once per beam (10x less K / V traffic at K = 10).  Opt-in: at the
# reference's eval sizes (469-token prompt, 10 beams) the decode step is bound by ~600 small launches, not by K / V bytes, and the
# grouped pass has 10x fewer workgroups in flight -- measured 6.84 vs 6.46 ms per token-step (profiles/r03_decode_timings.txt);
# it pays with long prompts / many users per call
DECODE_SHARED_PREFIX = _os_env("UNIMP_DECODE_SHARED_PREFIX", "0") != "0"


def attn_decode(q, k, v, scale, kv_len=None, alibi=None, out=None, group=1, shared_len=None):
    """one query row per (row, head): q [B,1,H,D] against the cached k / v [B,Sk,H,D] (strided views), kv_len int32 [B] keys
    visible per row (None: all Sk).  Split-key kernel (csrc/decode_attn.hip); returns o [B,1,H,D].
    group > 1 with shared_len (int32 [B // group]): the rows of a group (the beams of a prompt) hold identical K / V below
    shared_len[g]; those keys are read once per group."""
    B, Sq, H, D = q.shape
    assert Sq == 1
    Sk = k.shape[1]
    if out is None:
        out = torch.empty((B, 1, H, D), dtype=bf16, device=q.device)
    d = AttnDesc()
    (qp, qs), (kp, ks), (vp, vs), (op, os_) = _view4(q), _view4(k), _view4(v), _view4(out)
    _fill_attn(d, qp, kp, vp, op, 0, B, H, 1, Sk, D, scale, MASK_NONE, kv_len, None, 0, qs, ks, vs, os_, alibi)
    L = _lib.lib()
    splits = L.unimp_attn_decode_splits(B, H, Sk)
    grouped = group > 1 and shared_len is not None and B % group == 0 and group <= 16
    nslots = 2 * splits if grouped else splits
    ws = torch.empty(B * H * nslots * (D + 2), dtype=torch.float32, device=q.device) if nslots > 1 else None
    if grouped:
        assert shared_len.dtype == torch.int32 and shared_len.numel() == B // group
        check(L.unimp_attn_decode_grouped(C.byref(d), _p(ws), splits, group, shared_len.data_ptr(), _stream()), "attn_decode")
    else:
        check(L.unimp_attn_decode(C.byref(d), _p(ws), splits, _stream()), "attn_decode")
    return out


def attn_rope_fusable(dq, dk, dv, rope_half, D, adjacent=False):
    """can unimp_attn_bwd apply the transpose rotation to dq / dk itself?  (second-generation kernels, 16-byte aligned views;
    adjacent: the pair-adjacent layout of the GEMM's rotary epilogue instead of the half-split one)"""
    if attn_generation() < 2 or rope_half <= 0 or rope_half % (4 if adjacent else 8) or 2 * rope_half > D or dq.shape[1] < 4:
        return False
    for t in (dq, dk, dv):
        if t.data_ptr() % 16 or any(s_ % 8 for s_ in t.stride()[:3]):
            return False
    return True


def attn_generation():
    return _lib.lib().unimp_attn_get_generation()


def attn_bwd(q, k, v, o, lse, do, dq, dk, dv, scale, mask_mode=MASK_NONE, kv_len=None, seg=None, seg_len=0, alibi=None, rope=None,
             q_rows=None, k_rows=None):
    """writes dq/dk/dv (strided [B,S,H,D] views, every element of the views is overwritten).
    rope = (cos, sin) fp32 [positions][half] tables: dq and dk leave the kernels already rotated back (the transpose of the
    forward's rotation) -- only when attn_rope_fusable(); otherwise the caller runs rope_(inverse=True) itself.
    q_rows / k_rows: as in attn_fwd (do, dq packed like q; dk, dv like k); only the rows of the sequences are written -- the caller
    zeroes the buffer rows behind them if anything contracts over all rows."""
    d = AttnDesc()
    B, Sq, Sk = _packed_dims(d, q, k, q_rows, k_rows, kv_len)
    if k_rows is not None:
        kv_len = k_rows.len
    H, D = q.shape[2], q.shape[3]
    delta = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device)
    (qp, qs), (kp, ks), (vp, vs), (op, os_) = _view4(q), _view4(k), _view4(v), _view4(o)
    _fill_attn(d, qp, kp, vp, op, lse.data_ptr(), B, H, Sq, Sk, D, scale, mask_mode, kv_len, seg, seg_len, qs, ks, vs, os_, alibi)
    d.d_o, (d.do_bs, d.do_ss, d.do_hs) = _view4(do)
    d.dq, (d.dq_bs, d.dq_ss, d.dq_hs) = _view4(dq)
    d.dk, (d.dk_bs, d.dk_ss, d.dk_hs) = _view4(dk)
    d.dv, (d.dv_bs, d.dv_ss, d.dv_hs) = _view4(dv)
    d.delta = delta.data_ptr()
    if rope is not None and not torch.is_tensor(rope[0]):      # (half, log2 base): adjacent-pair layout, cos / sin computed in the kernels
        d.rope_half, d.rope_log2_base = int(rope[0]), float(rope[1])
    elif rope is not None:
        cos, sin = rope
        assert cos.dtype == torch.float32 and sin.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous()
        assert cos.shape == sin.shape and cos.shape[0] >= max(Sq, Sk), (cos.shape, Sq, Sk)
        d.rope_cos, d.rope_sin, d.rope_half = cos.data_ptr(), sin.data_ptr(), cos.shape[1]
    # a data-parallel group's collectives share the CUs with backward: attention3.hip's dK/dV kernel is one persistent workgroup per CU
    # (a CU held by a collective would leave its workgroup waiting for a free one: the kernel runs twice as long) -- like the persistent
    # GEMM variants it stays out while the flag is set.  Per call, in the descriptor (ABI 8): no process-wide library state is flipped.
    d.flags = 1 if AVOID_PERSISTENT else 0           # UNIMP_ATTN_NO_PERSISTENT
    check(_lib.lib().unimp_attn_bwd(C.byref(d), _stream()), "attn_bwd")


def embedding_fwd(ids, W, pos=None, P=None):
    rows = ids.numel()
    D = W.shape[1]
    out = torch.empty((rows, D), dtype=bf16, device=W.device)
    check(_lib.lib().unimp_embedding_fwd(_dev(ids).data_ptr(), W.data_ptr(), W.stride(0), _p(pos), _p(P),
                                          P.stride(0) if P is not None else 0, out.data_ptr(), D, rows, D, W.shape[0],
                                          _stream()), "embedding_fwd")
    return out


def embedding_bwd(ids, dout, vocab):
    """dense bf16 [vocab, D] gradient of the embedding table (fp32 sums in row order, then cast).  Bit-reproducible: the ids are sorted (stable)
    and every table row is summed by one writer (csrc/elementwise.hip embedding_bwd_sorted_kernel); the fp32-atomic form it replaces gave other
    last bits in 3 of 20 launches at the cfg2 shape (tools/check_embedding_bwd_determinism.py)."""
    rows, D = dout.shape
    acc = torch.zeros((vocab, D), dtype=torch.float32, device=dout.device)
    sid, perm = torch.sort(ids.reshape(-1), stable=True)
    L = _lib.lib()
    scratch = torch.empty(max(1, L.unimp_embedding_bwd_sorted_scratch(rows, D)), dtype=torch.float32, device=dout.device)
    check(L.unimp_embedding_bwd_sorted(sid.data_ptr(), perm.data_ptr(), dout.data_ptr(), dout.stride(0), acc.data_ptr(), D, scratch.data_ptr(), rows, D,
                                       vocab, _stream()), "embedding_bwd")
    return cast_bf16(acc)


def cast_bf16(src, scale=1.0):
    dst = torch.empty(src.shape, dtype=bf16, device=src.device)
    check(_lib.lib().unimp_cast_f32_to_bf16(_dev(src).data_ptr(), dst.data_ptr(), src.numel(), scale, _stream()), "cast")
    return dst


def add(a, b, out=None):
    assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
    if out is None:
        out = torch.empty_like(a)
    check(_lib.lib().unimp_add_bf16(_dev(a).data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "add")
    return out


def gather_rows(src, idx, out=None):
    """out[r] = src[idx[r]] (zeros where idx[r] < 0); src [rows_src, D] bf16 with unit inner stride, idx int32 [n]."""
    src, ld = _mat(src)
    n, D = idx.numel(), src.shape[1]
    assert idx.dtype == torch.int32 and idx.is_contiguous()
    if out is None:
        out = torch.empty((n, D), dtype=bf16, device=src.device)
    check(_lib.lib().unimp_gather_rows(src.data_ptr(), ld, _dev(idx).data_ptr(), out.data_ptr(), out.stride(0), n, D, _stream()), "gather_rows")
    return out


def marker(i):
    """empty kernel with grid = i workgroups on the current stream: a cut point for tools/trace_window.py"""
    check(_lib.lib().unimp_marker(int(i), _stream()), "marker")


def dot(a, b):
    """fp32 scalar tensor sum(a*b)."""
    assert a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel()
    out = torch.zeros(1 + 1024, dtype=torch.float32, device=a.device)      # [0]: result, [1:]: scratch of the ordered reduction
    check(_lib.lib().unimp_dot_bf16(_dev(a).data_ptr(), b.data_ptr(), a.numel(), out.data_ptr(), _stream()), "dot")
    return out[:1]


def vit_patchify(pixels, P, ldc):
    N, Cc, Hi, Wi = pixels.shape
    assert Cc == 3 and pixels.is_contiguous() and pixels.dtype in (torch.float32, bf16)
    g = (Hi // P) * (Wi // P)
    cols = torch.empty((N * g, ldc), dtype=bf16, device=pixels.device)
    check(_lib.lib().unimp_vit_patchify(_dev(pixels).data_ptr(), int(pixels.dtype == torch.float32), cols.data_ptr(), ldc, N, Hi,
                                         Wi, P, _stream()), "patchify")
    return cols


def vit_assemble(patch, cls, pos, N, n_patch):
    D = patch.shape[1]
    x = torch.empty((N, n_patch + 1, D), dtype=bf16, device=patch.device)
    check(_lib.lib().unimp_vit_assemble(_dev(patch).data_ptr(), patch.stride(0), cls.data_ptr(), pos.data_ptr(), x.data_ptr(), N,
                                         n_patch, D, _stream()), "vit_assemble")
    return x


def swiglu_fwd(gu, F):
    rows = gu.shape[0]
    out = torch.empty((rows, F), dtype=bf16, device=gu.device)
    check(_lib.lib().unimp_swiglu_fwd(_dev(gu).data_ptr(), gu.stride(0), out.data_ptr(), F, rows, F, _stream()), "swiglu_fwd")
    return out


def swiglu_bwd(gu, dout, F):
    rows = gu.shape[0]
    dgu = torch.empty((rows, 2 * F), dtype=bf16, device=gu.device)
    check(_lib.lib().unimp_swiglu_bwd(_dev(gu).data_ptr(), gu.stride(0), dout.data_ptr(), dout.stride(0), dgu.data_ptr(), 2 * F,
                                       rows, F, _stream()), "swiglu_bwd")
    return dgu


def bcast_rows(src, rows):
    period, D = src.shape
    out = torch.empty((rows, D), dtype=bf16, device=src.device)
    check(_lib.lib().unimp_bcast_rows(_dev(src).data_ptr(), out.data_ptr(), D, rows, period, D, _stream()), "bcast_rows")
    return out


def reduce_rows_periodic(src, period):
    rows, D = src.shape
    out = torch.empty((period, D), dtype=bf16, device=src.device)
    check(_lib.lib().unimp_reduce_rows_periodic(_dev(src).data_ptr(), src.stride(0), out.data_ptr(), rows, period, D, _stream()),
          "reduce_rows_periodic")
    return out


def label_mask(ids, answer_id, eoc_id, pad_id, media_id, want_labels=True, want_media_time=True):
    B, L = ids.shape
    ids = _dev(ids)
    if ids.dtype != torch.int64 or not ids.is_contiguous():        # the kernel reads contiguous int64 (collate_fn's dtype)
        ids = ids.long().contiguous()
    labels = torch.empty_like(ids) if want_labels else None
    mt = torch.empty((B, L), dtype=torch.int32, device=ids.device) if want_media_time else None
    check(_lib.lib().unimp_label_mask(ids.data_ptr(), _p(labels), _p(mt), B, L, answer_id, eoc_id, pad_id, media_id, _stream()),
          "label_mask")
    return labels, mt


def _focal_args(logits, labels, weights):
    """the kernels read bf16 logits with unit inner stride, contiguous int64 labels [B, L] and contiguous fp32 weights [B]:
    anything else (a batch moved with ``.to(device, dtype=bf16)``, double weights, int32 labels) is converted here."""
    if logits.dtype != bf16 or logits.stride(-1) != 1:
        raise _lib.UnimpHipError(f"focal_ce: logits must be bf16 with unit inner stride, got {logits.dtype} {tuple(logits.stride())}")
    B, L = logits.shape[0], logits.shape[1]
    if labels.shape != (B, L) or weights.numel() != B:
        raise _lib.UnimpHipError(f"focal_ce: labels {tuple(labels.shape)} / weights {tuple(weights.shape)} do not match logits {tuple(logits.shape)}")
    labels = _dev(labels)
    if labels.dtype != torch.int64 or not labels.is_contiguous():
        labels = labels.long().contiguous()
    weights = _dev(weights)
    if weights.dtype != torch.float32 or not weights.is_contiguous():
        weights = weights.float().contiguous()
    return labels, weights


def focal_ce_fwd(logits, V, labels, weights, gamma, use_reweight):
    """logits [B,L,ldv] bf16 (ldv >= V).  returns (row_lse, row_zy, out3 = [loss_sum, n_valid, ce_sum])."""
    labels, weights = _focal_args(logits, labels, weights)
    B, L, ldv = logits.shape[0], logits.shape[1], logits.stride(1)
    lse = torch.empty(B * L, dtype=torch.float32, device=logits.device)
    zy = torch.empty(B * L, dtype=torch.float32, device=logits.device)
    out3 = torch.empty(3, dtype=torch.float32, device=logits.device)
    check(_lib.lib().unimp_focal_ce_fwd(_dev(logits).data_ptr(), ldv, labels.data_ptr(), weights.data_ptr(), gamma, int(use_reweight),
                                         lse.data_ptr(), zy.data_ptr(), out3.data_ptr(), B, L, V, _stream()), "focal_ce_fwd")
    return lse, zy, out3


def focal_ce_bwd(logits, V, labels, weights, gamma, use_reweight, lse, zy, out3, gscale, dlogits):
    labels, weights = _focal_args(logits, labels, weights)
    B, L, ldv = logits.shape[0], logits.shape[1], logits.stride(1)
    assert dlogits.stride(1) == ldv
    check(_lib.lib().unimp_focal_ce_bwd(logits.data_ptr(), ldv, labels.data_ptr(), weights.data_ptr(), gamma, int(use_reweight),
                                         lse.data_ptr(), zy.data_ptr(), out3.data_ptr(), _p(gscale), dlogits.data_ptr(), B, L, V,
                                         _stream()), "focal_ce_bwd")
    return dlogits


def focal_ce_bwd_rows(logits, V, labels, weights, gamma, use_reweight, lse, zy, out3, gscale, rows):
    """compact gradient [n, ldd] (ldd = roundup8(V)) of the scored positions rows[i] = b*L + j; see include/unimp_hip.h."""
    labels, weights = _focal_args(logits, labels, weights)
    L, ldv = logits.shape[1], logits.stride(1)
    assert rows.dtype == torch.int64 and rows.is_contiguous()
    n, ldd = rows.numel(), (V + 7) // 8 * 8
    buf = torch.empty((n, ldd), dtype=bf16, device=logits.device)
    check(_lib.lib().unimp_focal_ce_bwd_rows(logits.data_ptr(), ldv, labels.data_ptr(), weights.data_ptr(), gamma, int(use_reweight),
                                              lse.data_ptr(), zy.data_ptr(), out3.data_ptr(), _p(gscale), _dev(rows).data_ptr(), n,
                                              buf.data_ptr(), ldd, L, V, _stream()), "focal_ce_bwd_rows")
    return buf[:, :V]


def sumsq(g, out):
    """out: fp32 [1 + 1024]; out[0] += sum(g^2)."""
    check(_lib.lib().unimp_sumsq_bf16(_dev(g).data_ptr(), g.numel(), out.data_ptr(), _stream()), "sumsq")


def adamw_flat(master, m, v, p16, g16, n_decay, lr, beta1, beta2, eps, wd, step, sumsq_buf, gscale, max_norm, zero_grad=True):
    check(_lib.lib().unimp_adamw_flat(_dev(master).data_ptr(), m.data_ptr(), v.data_ptr(), p16.data_ptr(), g16.data_ptr(),
                                       master.numel(), n_decay, lr, beta1, beta2, eps, wd, step, _p(sumsq_buf), gscale, max_norm,
                                       int(zero_grad), _stream()), "adamw")


# ------------------------------------------------------------------------------------------------- MX-fp8 (frozen towers, F4)
class MxTensor:
    """OCP e4m3 elements [R, K] (uint8) + one E8M0 scale byte per 32 consecutive k: scales [R, K / 32] (uint8)."""
    __slots__ = ("q", "scales", "K")

    def __init__(self, q, scales):
        self.q, self.scales, self.K = q, scales, q.shape[1]

    def dequantize(self):
        """fp32 [R, K]: element * 2^(scale - 127) (tests; the GEMM never materialises this)."""
        e = self.q.view(torch.float8_e4m3fn).float()
        s = torch.pow(2.0, self.scales.float() - 127.0).repeat_interleave(32, dim=1)
        return e * s


def mx_quantize(x):
    """bf16 [R, K] (unit inner stride, K % 32 == 0) -> MxTensor, quantised along K (include/unimp_hip.h)."""
    x, ldx = _mat(x)
    R, K = x.shape
    q = torch.empty((R, K), dtype=torch.uint8, device=x.device)
    sc = torch.empty((R, K // 32), dtype=torch.uint8, device=x.device)
    check(_lib.lib().unimp_mx_quantize(x.data_ptr(), ldx, q.data_ptr(), K, sc.data_ptr(), K // 32, R, K, _stream()), "mx_quantize")
    return MxTensor(q, sc)


def gemm_mx(a, b, *, bias=None, act=None, pre=None, aux=None, res=None, out=None, out_mx=False):
    """C[M, N] (bf16) = epi(A B^T) for MxTensors a [M, K], b [N, K]; epilogue as unimp_gemm_mxfp8.  A uint8 ``pre`` / ``aux`` is the
    8-bit stored derivative of the bf16 GEMMs (act'(z), step 1 / 202); they cannot be mixed with bf16 ones in one call.
    ``out_mx``: return the result as an MxTensor quantised along N (the operand of the next product) -- the same bytes as
    ``mx_quantize(gemm_mx(...))`` without the bf16 tensor and the quantiser pass (N % 32 == 0)."""
    M, N, K = a.q.shape[0], b.q.shape[0], a.K
    assert b.K == K, (a.q.shape, b.q.shape)
    sc_out = None
    if out_mx:
        assert out is None and N % 32 == 0, (N,)
        out = torch.empty((M, N), dtype=torch.uint8, device=a.q.device)
        sc_out = torch.empty((M, N // 32), dtype=torch.uint8, device=a.q.device)
    elif out is None:
        out = torch.empty((M, N), dtype=bf16, device=a.q.device)
    d = MxGemmDesc()
    if sc_out is not None:
        d.scale_c, d.ldsc = sc_out.data_ptr(), sc_out.stride(0)
    d.A, d.B, d.scale_a, d.scale_b, d.C = a.q.data_ptr(), b.q.data_ptr(), a.scales.data_ptr(), b.scales.data_ptr(), _dev(out).data_ptr()
    d.lda, d.ldb, d.ldsa, d.ldsb, d.ldc = a.q.stride(0), b.q.stride(0), a.scales.stride(0), b.scales.stride(0), out.stride(0)
    d.bias = _p(bias)
    if res is not None:
        d.res, d.ldres = res.data_ptr(), res.stride(0)
    if aux is not None:
        d.aux, d.ldaux = aux.data_ptr(), aux.stride(0)
    if pre is not None:
        d.pre, d.ldpre = pre.data_ptr(), pre.stride(0)
    d.M, d.N, d.K, d.act = M, N, K, ACT[act]
    u8 = [t.dtype == torch.uint8 for t in (pre, aux) if t is not None]
    assert all(u8) or not any(u8), "gemm_mx: pre and aux must both be uint8 or both bf16"
    d.deriv_u8 = int(bool(u8) and u8[0])
    if GEMM_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_lib.lib().unimp_gemm_mxfp8(C.byref(d), _stream()), "gemm_mxfp8")
    if GEMM_PROFILE is not None:
        e1.record()
        epi = "+".join(n for n, t in (("bias", bias), ("act", act), ("pre", pre), ("aux", aux), ("res", res)) if t is not None) or "plain"
        GEMM_PROFILE.append((e0, e1, 2.0 * M * N * K, (M, N, K, epi + ("->mx" if out_mx else ""), 0, "mxfp8")))
    return MxTensor(out, sc_out) if out_mx else out
