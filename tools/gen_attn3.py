#!/usr/bin/env python3
"""Generates unimp_amd/csrc/attention3_sched.inc: the instruction schedule of one query tile of the third-generation dK/dV kernel
(attention3.hip) as a sequence of `asm volatile` statements, ONE per matrix instruction: the MFMA first, then the vector / LDS
instructions that issue while it runs ("fillers").  hipcc keeps volatile asm statements in source order and models none of their
contents, so everything a schedule must respect is checked HERE:

  * an LDS read's destination is consumed only after a counted `s_waitcnt lgkmcnt(N)` that the generator derives from the issue
    order (N = reads issued after the needed one, capped at 15; LDS returns in order, the loop has no scalar load);
  * a vector instruction reads an MFMA result no earlier than two MFMA slots later (8-pass XDL write -> VALU read: 12 wait
    states), an MFMA reads a vector result no earlier than the slot after next (2 wait states), a `v_cvt_pk` reads a `v_exp`
    result with at least one instruction between (gfx940 trans forwarding);
  * per slot the issue cost (MFMA 8, v_exp 8, other vector / LDS 4 cycles: MI355X_MICROARCH.md 'vector-instruction ISSUE cost')
    is reported; the 32-cycle slot of a v_mfma_f32_32x32x16_bf16 hides 24.

Three variants of the tile (both 32-key blocks of the wave visible, only the first, only the second); masked blocks differ only
in the per-row exponent offsets the C++ side passes (a masked element's offset is -1e30: p = 0).

    python tools/gen_attn3.py            # rewrites the .inc
    python tools/gen_attn3.py --report   # per-slot cost table on stdout
    python tools/gen_attn3.py --check    # the committed .inc is what this script generates (tests/test_cabi_cpu.py)
"""
import argparse
import os
from collections import OrderedDict

KS, ND = 5, 3
OFF_DO = 6144
ROW16 = 16 * 12 * 16          # bytes of 16 image rows


class Stmt:
    """one asm statement: instruction templates with symbolic operands"""

    def __init__(self):
        self.lines = []
        self.ops = OrderedDict()     # expr -> [mode, cls]; mode: 'in', 'out', 'early', 'inout'
        self.cost = 0
        self.tags = []

    def op(self, expr, mode, cls="v"):
        if expr in self.ops:
            old = self.ops[expr][0]
            if old != mode:
                # a value written earlier in the statement and read later (or the reverse): read-write.  An 'early' (MFMA result)
                # never meets another use inside one statement by construction.
                assert "early" not in (old, mode), (expr, old, mode)
                self.ops[expr][0] = "inout" if old in ("in", "inout") or mode in ("in", "inout") else "out"
        else:
            self.ops[expr] = [mode, cls]
        return "{" + expr + "}"

    def add(self, text, cost, tag=None):
        self.lines.append(text)
        self.cost += cost
        if tag:
            self.tags.append(tag)

    def render(self, indent="    "):
        outs = [(e, m, c) for e, (m, c) in self.ops.items() if m in ("out", "early", "inout")]
        ins = [(e, m, c) for e, (m, c) in self.ops.items() if m == "in"]
        index = {}
        for i, (e, _, _) in enumerate(outs + ins):
            index[e] = i
        assert len(index) <= 30, "clang: at most 30 asm operands"
        body = []
        for ln in self.lines:
            for e, i in index.items():
                ln = ln.replace("{" + e + "}", "%" + str(i))
            body.append(ln)
        # every output of a multi-instruction statement is early-clobber: hipcc may otherwise give it the register of an input whose
        # last use is this statement -- and an instruction further down the string still reads that input
        multi = len(self.lines) > 1
        def cons(m, c):
            return {"out": ("=&" if multi else "=") + c, "early": "=&" + c, "inout": "+" + c, "in": c}[m]
        o = ", ".join('"%s"(%s)' % (cons(m, c), e) for e, m, c in outs)
        i_ = ", ".join('"%s"(%s)' % (cons(m, c), e) for e, m, c in ins)
        text = "\\n\\t".join(body)
        return '%sasm volatile("%s" : %s : %s);' % (indent, text, o, i_)


def sreg(kb):
    return "s%d" % kb


def dpreg(kb):
    return "dp%d" % kb


class Sched:
    def __init__(self, vis):
        self.vis = vis                 # (kb0 visible, kb1 visible)
        self.kbs = [kb for kb in (0, 1) if vis[kb]]
        self.slots = []                # list of Stmt
        self.reads = []                # issue order of LDS reads: names
        self.read_slot = {}
        self.mfma_slot = {}            # (kind, ...) -> slot
        self.done_slot = {}            # filler name -> slot
        self.pre = []                  # statements before the first slot (reads issued ahead)

    # ---- LDS reads (may be placed in the preamble or as fillers)
    def rd_row(self, st, name, k, base_off):
        addr = "a_rb" if (k & 1) else "a_ra"
        st.add("ds_read_b128 %s, %s offset:%d" % (st.op("%s[%d]" % (name, k), "out"), st.op(addr, "in"), base_off + (k >> 1) * 64), 4)
        self.reads.append("%s%d" % (name, k))

    def rd_tr(self, st, name, i, base_off):
        nd, kk = i // 2, i % 2
        off = base_off + kk * ROW16 + nd * 64
        st.add("ds_read_b64_tr_b16 %s, %s offset:%d" % (st.op("%sl[%d]" % (name, i), "out"), st.op("a_t0", "in"), off), 4)
        st.add("ds_read_b64_tr_b16 %s, %s offset:%d" % (st.op("%sh[%d]" % (name, i), "out"), st.op("a_t1", "in"), off), 4)
        self.reads.append("%sl%d" % (name, i))
        self.reads.append("%sh%d" % (name, i))

    def wait_for(self, st, last_read, exprs):
        """counted wait until `last_read` (a name in self.reads) has landed; names the released registers"""
        after = len(self.reads) - 1 - self.reads.index(last_read)
        n = min(after, 15)
        for e in exprs:
            st.op(e, "inout")
        st.add("s_waitcnt lgkmcnt(%d)" % n, 0)

    # ---- fillers (one element or one packed pair each, so that the list scheduler can fill a slot to its 32 cycles)
    def f_ex(self, st, kb, r):
        a = "%s[%d]" % (sreg(kb), r)
        st.add("v_fma_f32 %s, %s, %s, %s" % (st.op(a, "inout"), st.op(a, "inout"), st.op("sc2", "in"), st.op("nl%d[%d]" % (kb, r), "in")), 4)
        st.add("v_exp_f32 %s, %s" % (st.op(a, "inout"), st.op(a, "inout")), 8)

    def f_pc(self, st, kb, j):
        s = sreg(kb)
        st.add("v_cvt_pk_bf16_f32 %s, %s, %s" % (st.op("pf[%d][%d][%d]" % (kb, j >> 2, j & 3), "out"),
                                                  st.op("%s[%d]" % (s, 2 * j), "in"), st.op("%s[%d]" % (s, 2 * j + 1), "in")), 4)

    def f_gm(self, st, kb, r):
        s, d = sreg(kb), dpreg(kb)
        st.add("v_mul_f32 %s, %s, %s" % (st.op("%s[%d]" % (d, r), "inout"), st.op("%s[%d]" % (s, r), "in"), st.op("%s[%d]" % (d, r), "inout")), 4)

    def f_dc(self, st, kb, j):
        d = dpreg(kb)
        st.add("v_cvt_pk_bf16_f32 %s, %s, %s" % (st.op("dsf[%d][%d][%d]" % (kb, j >> 2, j & 3), "out"),
                                                  st.op("%s[%d]" % (d, 2 * j), "in"), st.op("%s[%d]" % (d, 2 * j + 1), "in")), 4)

    # ---- MFMAs
    def m_s(self, st, ks, kb):
        if ks == 0:
            st.add("v_mfma_f32_32x32x16_bf16 %s, %s, %s, 0" % (st.op(sreg(kb), "early"), st.op("qf[%d]" % ks, "in"), st.op("kf[%d][%d]" % (kb, ks), "in")), 8)
        else:
            st.add("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (st.op(sreg(kb), "inout"), st.op("qf[%d]" % ks, "in"), st.op("kf[%d][%d]" % (kb, ks), "in"),
                                                                 st.op(sreg(kb), "inout")), 8)

    def m_d(self, st, ks, kb):
        if ks == 0:
            st.add("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (st.op(dpreg(kb), "early"), st.op("dof[%d]" % ks, "in"), st.op("vf[%d][%d]" % (kb, ks), "in", "a"),
                                                                 st.op("ndl", "inout")), 8)      # read-write: no output of this statement may be allocated over the C operand the MFMA is still reading
        else:
            st.add("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (st.op(dpreg(kb), "inout"), st.op("dof[%d]" % ks, "in"), st.op("vf[%d][%d]" % (kb, ks), "in", "a"),
                                                                 st.op(dpreg(kb), "inout")), 8)

    def m_acc(self, st, acc, kb, nd, kk, frag, b):
        a = "%s[%d][%d]" % (acc, kb, nd)
        st.add("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (st.op(a, "inout", "a"), st.op("a3_join(%sl[%d], %sh[%d])" % (frag, 2 * nd + kk, frag, 2 * nd + kk), "in"),
                                                             st.op("%s[%d][%d]" % (b, kb, kk), "in"), st.op(a, "inout", "a")), 8)

    # ---- the schedule
    def build(self):
        kbs = self.kbs
        mf = []                        # MFMA sequence
        # block by block (not interleaved by k-step): the first block's S is complete five slots earlier and its exponentials
        # start under the second block's S; the Q / dO fragments stay in registers for the second block (2 x 20)
        for kb in kbs:
            for ks in range(KS):
                mf.append(("S", ks, kb))
        for ks in range(KS):
            for kb in kbs:
                mf.append(("D", ks, kb))          # dP: k-step by k-step (a dO fragment serves both blocks and dies; 12 registers less at the peak)
        order = [(nd, kk) for kk in range(2) for nd in range(ND)]          # both accumulate chains of a block start on P / dS dwords 0-3
        # dV of both blocks, then dK of both: the dO^T fragments (24 registers) are dead before the Q^T fragments arrive
        for kb in kbs:
            for nd, kk in order:
                mf.append(("A", kb, nd, kk))
        for kb in kbs:
            for nd, kk in order:
                mf.append(("B", kb, nd, kk))
        slot_of = {m: i for i, m in enumerate(mf)}
        last_s = {kb: slot_of[("S", KS - 1, kb)] for kb in kbs}
        last_d = {kb: slot_of[("D", KS - 1, kb)] for kb in kbs}
        first_a = {kb: slot_of[("A", kb, 0, 0)] for kb in kbs}
        first_b = {kb: slot_of[("B", kb, 0, 0)] for kb in kbs}

        # vector work units: name -> dict(cost, fn, deadline, deps (units that must sit in an EARLIER slot), after (same slot allowed, earlier in order), ready)
        units = OrderedDict()
        for kb in kbs:
            for j in range(8):
                dl_pc = first_a[kb] + 3 * (j >> 2) - 2          # an MFMA reads a vector result two slots later at the earliest
                for r in (2 * j, 2 * j + 1):
                    units["EX%d_%d" % (kb, r)] = dict(cost=12, fn=(lambda st, kb=kb, r=r: self.f_ex(st, kb, r)), deadline=dl_pc - 1, deps=[], after=[],
                                                     ready=last_s[kb] + 2)
                units["PC%d_%d" % (kb, j)] = dict(cost=4, fn=(lambda st, kb=kb, j=j: self.f_pc(st, kb, j)), deadline=dl_pc,
                                                 deps=["EX%d_%d" % (kb, 2 * j), "EX%d_%d" % (kb, 2 * j + 1)], after=[], ready=0)
            for j in range(8):
                dl_dc = first_b[kb] + 3 * (j >> 2) - 2
                for r in (2 * j, 2 * j + 1):
                    units["GM%d_%d" % (kb, r)] = dict(cost=4, fn=(lambda st, kb=kb, r=r: self.f_gm(st, kb, r)), deadline=dl_dc, deps=["EX%d_%d" % (kb, r)], after=[],
                                                     ready=last_d[kb] + 2)
                units["DC%d_%d" % (kb, j)] = dict(cost=4, fn=(lambda st, kb=kb, j=j: self.f_dc(st, kb, j)), deadline=dl_dc, deps=[],
                                                 after=["GM%d_%d" % (kb, 2 * j), "GM%d_%d" % (kb, 2 * j + 1)], ready=0)
        # the packed P of pair j must be formed before the dS multiply overwrites nothing of s (GM reads s, writes dp): no conflict
        pend = list(units.keys())
        pend.sort(key=lambda n: units[n]["deadline"])          # stable: list order inside one deadline

        # LDS reads: Q fragments ahead of the first slot, then one unit per slot in use order
        pre = Stmt()
        for k in range(KS):
            self.rd_row(pre, "qf", k, 0)
        self.pre.append(pre)
        rd_queue = [("dof", k) for k in range(KS)] + [("ud", 2 * nd + kk) for nd, kk in order] + [("uq", 2 * nd + kk) for nd, kk in order]

        # dO fragments are needed from the first D slot on, the transposed ones from the first A / B slot: one unit per slot, late
        # enough that no fragment waits in a register for longer than it must
        rd_start = max(0, slot_of[("D", 0, kbs[0])] - 6)
        # the Q^T fragments (dK, the last phase) are read as late as their first use allows: they take over the registers of the dO^T
        # fragments as those die (both sets live at once cost 48 registers at the kernel's peak)
        uq_start = max(0, first_b[kbs[0]] - 6)
        # one counted wait per GROUP of fragments (every wait statement is an issue slot, and hipcc pads the statement after it): all
        # Q fragments before the first S, all dO fragments before the first dP; the transposed fragments in two halves (before the
        # first and before the fourth MFMA of their phase: in the one-block variants the second half is still being read at the first)
        need = {}
        first_s, first_d = slot_of[("S", 0, kbs[0])], slot_of[("D", 0, kbs[0])]
        fa, fb = first_a[kbs[0]], first_b[kbs[0]]
        need[first_s] = ("qf%d" % (KS - 1), ["qf[%d]" % k for k in range(KS)])
        need[first_d] = ("dof%d" % (KS - 1), ["dof[%d]" % k for k in range(KS)])
        fr = [2 * nd + kk for nd, kk in order]
        need[fa] = ("udh%d" % fr[2], ["ud%s[%d]" % (hl, f) for f in fr[:3] for hl in "lh"])
        need[fa + 3] = ("udh%d" % fr[-1], ["ud%s[%d]" % (hl, f) for f in fr[3:] for hl in "lh"])
        need[fb] = ("uqh%d" % fr[2], ["uq%s[%d]" % (hl, f) for f in fr[:3] for hl in "lh"])
        need[fb + 3] = ("uqh%d" % fr[-1], ["uq%s[%d]" % (hl, f) for f in fr[3:] for hl in "lh"])
        released = set()
        placed = {}
        self.waits = {}
        for i, m in enumerate(mf):
            st = Stmt()
            nm, exprs = need.get(i, (None, None))
            if nm is not None and nm not in released:
                assert nm in self.reads, "read %s not issued before slot %d" % (nm, i)
                w = Stmt()
                self.wait_for(w, nm, exprs)
                self.waits[i] = w
                for r in self.reads[: self.reads.index(nm) + 1]:
                    released.add(r)
            if m[0] == "S":
                self.m_s(st, m[1], m[2])
            elif m[0] == "D":
                self.m_d(st, m[1], m[2])
            elif m[0] == "A":
                self.m_acc(st, "dv", m[1], m[2], m[3], "ud", "pf")
            else:
                self.m_acc(st, "dk", m[1], m[2], m[3], "uq", "dsf")
            if rd_queue and i >= (uq_start if rd_queue[0][0] == "uq" else rd_start):
                kind, k = rd_queue.pop(0)
                if kind == "dof":
                    self.rd_row(st, "dof", k, OFF_DO)
                elif kind == "ud":
                    self.rd_tr(st, "ud", k, OFF_DO)
                else:
                    self.rd_tr(st, "uq", k, 0)
            def can(nme):
                u = units[nme]
                if u["ready"] > i:
                    return False
                if any(placed.get(d, 1 << 30) >= i for d in u["deps"]):
                    return False
                if any(d not in placed for d in u["after"]):
                    return False
                return True
            progress = True
            while progress:
                progress = False
                for nme in list(pend):
                    u = units[nme]
                    if not can(nme):
                        continue
                    forced = u["deadline"] <= i
                    if forced or st.cost + u["cost"] <= 32:
                        u["fn"](st)
                        placed[nme] = i
                        pend.remove(nme)
                        progress = True
                        break
            late = [nme for nme in pend if units[nme]["deadline"] <= i]
            assert not late, ("cannot meet", late, "at slot", i)
            self.slots.append(st)
        assert not pend, pend
        # no statement may name an accumulator both whole and by element
        for st in self.slots:
            for x in ("s0", "s1", "dp0", "dp1"):
                assert not (x in st.ops and any(e.startswith(x + "[") for e in st.ops)), (x, list(st.ops))
        self.mf = mf
        self.placed = placed
        return self

    def render(self, indent="    "):
        out = []
        for st in self.pre:
            out.append(st.render(indent))
        for i, st in enumerate(self.slots):
            out.append("%s// slot %d: %s  (issue cost %d)" % (indent, i, " ".join(str(x) for x in self.mf[i]), st.cost))
            if i in self.waits:
                out.append(self.waits[i].render(indent))
            out.append(st.render(indent))
        return "\n".join(out)


HEADER = """// GENERATED by tools/gen_attn3.py -- do not edit; the schedule and its checks live there.
// One query tile of attn_dkv3_kernel (attention3.hip): every statement is one v_mfma_f32_32x32x16_bf16 followed by the vector and
// LDS instructions that issue while it runs.  Names in scope: s0 s1 dp0 dp1 nl0 nl1 ndl (f32x16), qf dof (bf16x8[5]),
// udl udh uql uqh (s16x4[6]), pf dsf (u32x4[2][2]), kf vf (bf16x8[2][5]), dk dv (f32x16[2][3], AGPRs), a_ra a_rb a_t0 a_t1 (LDS byte
// addresses inside the stage), sc2.
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--report", action="store_true")
    ap.add_argument("--check", action="store_true", help="do not write: exit 1 if the committed .inc differs from what this script generates")
    args = ap.parse_args()
    variants = [((True, True), "V0 && V1"), ((True, False), "V0"), ((False, True), "true")]
    text = [HEADER]
    for n, (vis, cond) in enumerate(variants):
        s = Sched(vis).build()
        kw = "if" if n == 0 else "else if"
        text.append("%s constexpr (%s) {" % (kw, cond) if cond != "true" else "else {")
        text.append(s.render())
        text.append("}")
        if args.report:
            print("variant", vis, "slots", len(s.slots), "total issue cost", sum(st.cost for st in s.slots))
            for i, st in enumerate(s.slots):
                print("  %2d %-12s cost %3d  %s" % (i, " ".join(str(x) for x in s.mf[i]), st.cost, " ".join(k for k, v in s.placed.items() if v == i)))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "unimp_amd", "csrc", "attention3_sched.inc")
    if args.check:
        same = os.path.exists(path) and open(path).read() == "\n".join(text) + "\n"
        print("attention3_sched.inc is", "up to date" if same else "STALE: run tools/gen_attn3.py")
        raise SystemExit(0 if same else 1)
    with open(path, "w") as f:
        f.write("\n".join(text) + "\n")


if __name__ == "__main__":
    main()
