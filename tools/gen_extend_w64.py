#!/usr/bin/env python3
"""Generator of scratchpad_amd/csrc/extend_w64_gen.inc: the register-literal parts of extend_w64.hip.

extend_w64.hip keeps O^T (8 blocks of 16) and the Q fragments (16 of 4) in accumulation registers that the ASSEMBLY
owns (a[0:127], a[128:191]): the compiler sees neither their values nor their lifetimes (handing them over as
operands - also pinned to physical registers - made it keep second copies at every control-flow join and spill), so
every instruction that touches them is `asm volatile` text with the register names spelled out, text that C++
templates cannot build; tests/test_extend_isa.py checks on the assembly that no compiler-made instruction touches an
accumulation register.  This script writes that text as macros:

  SP_W64_INIT_O / SP_W64_SET_Q(qf)          zero O^T, move the Q fragments in
  SP_W64_GEN_MFMA_S(PAR, KB, kfr)           the 16 S^T MFMAs of one 32-key block (compiler-scheduled callers)
  SP_W64_GEN_PV_STEP(KS)                    the 8 O^T MFMAs of one 16-key k-step
  SP_W64_RESCALE_O(RB, al)                  O^T rows of a row block times a per-lane factor (rare path)
  SP_W64_READ_O(RB, DB, x)                  16 accumulators into float x[16] (epilogue)
  SP_W64_STEADY_0..3                        the software-pipelined tile iteration, one per ring buffer, every MFMA gap
                                            filled by the table below, with counted lgkmcnt waits

Usage: python tools/gen_extend_w64.py [--check]   (--check: exit 1 if the committed file differs)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "scratchpad_amd", "csrc", "extend_w64_gen.inc")

K_TILE = 16384          # bytes of a K (or V) tile in LDS
ROW_B = 256


def oreg(rb, db):
    n = (rb * 4 + db) * 16
    return f"a[{n}:{n + 15}]"


def qreg(rb, ks):
    n = 128 + (rb * 8 + ks) * 4
    return f"a[{n}:{n + 3}]"


def clobbers(lo, hi):
    return ", ".join(f'"a{i}"' for i in range(lo, hi))


def sreg(par, e):
    """register of exponential slot e (0..63) of the tile with parity par: k-step e >> 4, row block (e >> 3) & 1"""
    return f"S[{par}][{e >> 5}][{(e >> 3) & 1}][{8 * ((e >> 4) & 1) + (e & 7)}]"


class Macro:
    def __init__(self, head):
        self.lines = [f"#define {head}"]

    def add(self, text=""):
        self.lines.append("  " + text)

    def render(self):
        body = self.lines
        width = 118
        out = []
        for i, ln in enumerate(body):
            out.append(ln + (" " * max(1, width - len(ln)) + "\\" if i + 1 < len(body) else ""))
        return "\n".join(out) + "\n"


def mfma(dst, a, b, c, operands):
    return f'SP_W64_MFMA("{dst}", "{a}", "{b}", "{c}", {operands});'


def mfma_s(par, kb, rb, ks, kfrag):
    """S[par][kb][rb] (+)= K fragment . Q[rb][ks] (the Q fragment by its register name)"""
    if ks == 0:
        return mfma("%0", "%1", qreg(rb, ks), "0", f': "=&v"(S[{par}][{kb}][{rb}]) : "v"({kfrag})')
    return mfma("%0", "%1", qreg(rb, ks), "%0", f': "+v"(S[{par}][{kb}][{rb}]) : "v"({kfrag})')


def mfma_o(rb, db, vfrag, pfrag):
    return mfma(oreg(rb, db), "%0", "%1", oreg(rb, db), f':: "v"({vfrag}), "v"({pfrag})')


def gen_static():
    out = []
    m = Macro("SP_W64_INIT_O")
    for rb in range(2):
        for db in range(4):
            n = (rb * 4 + db) * 16
            txt = " ".join(f'"v_accvgpr_write_b32 a{n + i}, 0\\n"' for i in range(16))
            m.add(f'asm volatile({txt} ::: {clobbers(n, n + 16)});')
    out.append(m.render())
    m = Macro("SP_W64_SET_Q(QF)")
    for rb in range(2):
        for ks in range(8):
            n = 128 + (rb * 8 + ks) * 4
            txt = " ".join(f'"v_accvgpr_write_b32 a{n + i}, %{i}\\n"' for i in range(4))
            ins = ", ".join(f'"v"(QF[{rb}][{ks}][{i}])' for i in range(4))
            m.add(f'asm volatile({txt} :: {ins} : {clobbers(n, n + 4)});')
    out.append(m.render())
    m = Macro("SP_W64_GEN_MFMA_S(PAR, KB, KFR)")
    for ks in range(8):
        for rb in range(2):
            m.add(mfma_s("PAR", "KB", rb, ks, f"KFR[{ks}]"))
    out.append(m.render())
    m = Macro("SP_W64_GEN_PV_STEP(KS)")
    for db in range(4):
        m.add(f"{{ u32x4 vf_; vf_[0] = vlo[(KS) & 1][{db}][0]; vf_[1] = vlo[(KS) & 1][{db}][1]; vf_[2] = vhi[(KS) & 1][{db}][0]; "
              f"vf_[3] = vhi[(KS) & 1][{db}][1];")
        for rb in range(2):
            m.add("  " + mfma_o(rb, db, "vf_", f"pf[{rb}][KS]"))
        m.add("}")
    out.append(m.render())
    for rb in range(2):
        m = Macro(f"SP_W64_RESCALE_O_{rb}(AL)")
        m.add("{ float t0_, t1_, t2_, t3_;")
        for db in range(4):
            n = (rb * 4 + db) * 16
            txt = []
            for base in range(n, n + 16, 4):
                txt += [f'"v_accvgpr_read_b32 %{i}, a{base + i}\\n"' for i in range(4)]
                txt += [f'"v_mul_f32 %{i}, %{i}, %4\\n"' for i in range(4)]
                txt += [f'"v_accvgpr_write_b32 a{base + i}, %{i}\\n"' for i in range(4)]
            m.add(f'asm volatile({" ".join(txt)} : "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_) : "v"(AL) : {clobbers(n, n + 16)});')
        m.add("}")
        out.append(m.render())
    for rb in range(2):
        for db in range(4):
            n = (rb * 4 + db) * 16
            m = Macro(f"SP_W64_READ_O_{rb}_{db}(X)")
            rd = " ".join(f'"v_accvgpr_read_b32 %{i}, a{n + i}\\n"' for i in range(16))
            outs = ", ".join(f'"=v"(X[{i}])' for i in range(16))
            m.add(f'asm volatile({rd} : {outs});')
            out.append(m.render())
    return "".join(out)


# ---------------------------------------------------------------------------------------------- the steady body
def slot_of(n):
    """exponential slot n of an iteration -> (which tile, e): slots 0..41 finish the tile in its late stage ('cur'),
    42..63 start the next one ('nxt'); -1 and 64 are the neighbours' edge slots"""
    if n < 0:
        return ("cur", 22 + n)
    if n <= 41:
        return ("cur", n + 22)
    return ("nxt", n - 42)


def steady(buf):
    P, Pn = buf & 1, (buf & 1) ^ 1
    par = {"cur": P, "nxt": Pn}
    buf1, buf2, bufd = (buf + 1) % 4, (buf + 2) % 4, (buf + 3) % 4
    # two index sets: the pieces of tile t+3 go out from the set loaded one iteration ago, the other one is refilled
    # with tile t+4's indices (an iteration is ~2,300 cycles, an index load well under 1,000)
    refill, issue = [("sA", "sB"), ("sB", "sA")][buf & 1]
    # ---- LDS reads by gap: ("K", frag, tile buffer, dest slot) / ("V", v, half)
    reads = {g: [] for g in range(64)}
    for f in range(2, 16):
        reads[2 * f - 4].append(("K", f, buf1))
    reads[60].append(("K", 0, buf2))
    reads[62].append(("K", 1, buf2))
    for v in range(16):
        reads[24 + 2 * v].append(("V", v, 0))
        reads[25 + 2 * v].append(("V", v, 1))
    # program order of LDS ops over one iteration (a gap's wait and MFMA come first, its reads after them)
    order = []
    for g in range(64):
        for r in reads[g]:
            order.append((g, r))

    m = Macro(f"SP_W64_STEADY_{buf}")
    m.add("{ float mxr_[2], mc_[2]; bool any_ = false;")
    for g in range(64):
        m.add(f"/* gap {g} */")
        # ---- the MFMA (and the counted wait for its fragment: issued by the first MFMA that uses it)
        if g < 32:
            kb, ks, rb = g >> 4, (g >> 1) & 7, g & 1
            f = kb * 8 + ks
            if rb == 0:
                m.add(f'asm volatile("s_waitcnt lgkmcnt({wait_count_k(order, g, f)})" : "+v"(kf[{f % 3}]));')
            m.add(mfma_s(Pn, kb, rb, ks, f"kf[{f % 3}]"))
        else:
            j = g - 32
            s, db, rb = j >> 3, (j >> 1) & 3, j & 1
            v = s * 4 + db
            if rb == 0:
                m.add(f'asm volatile("s_waitcnt lgkmcnt({wait_count_v(order, g, v)})" : "+v"(vlo[{s & 1}][{db}]), "+v"(vhi[{s & 1}][{db}]));')
            m.add(f"{{ u32x4 vf_; vf_[0] = vlo[{s & 1}][{db}][0]; vf_[1] = vlo[{s & 1}][{db}][1]; vf_[2] = vhi[{s & 1}][{db}][0]; "
                  f"vf_[3] = vhi[{s & 1}][{db}][1];")
            m.add("  " + mfma_o(rb, db, "vf_", f"pf[{rb}][{s}]") + " }")
        # ---- LDS reads
        for r in reads[g]:
            if r[0] == "K":
                _, f, tb = r
                kb, ks = f >> 3, f & 7
                off = tb * K_TILE + kb * 32 * ROW_B
                m.add(f'asm volatile("ds_read_b128 %0, %1 offset:{off}" : "=v"(kf[{f % 3}]) : "v"(kaddr[{ks}]));')
            else:
                _, v, half = r
                s, db = v >> 2, v & 3
                off = buf * K_TILE + s * 16 * ROW_B + half * 8 * ROW_B
                dst = f"vhi[{s & 1}][{db}]" if half else f"vlo[{s & 1}][{db}]"
                m.add(f'asm volatile("ds_read_b64_tr_b16 %0, %1 offset:{off}" : "=v"({dst}) : "v"(va[{db}]));')
        # ---- exponential pipeline: fma of slot g+1, exp of slot g, add / pack of slot g-1
        who, e = ("nxt", 22) if g == 63 else slot_of(g + 1)
        m.add(f"{sreg(par[who], e)} = __builtin_fmaf({sreg(par[who], e)}, sc, negm[{par[who]}][{(e >> 3) & 1}]);")
        who, e = slot_of(g)
        m.add(f"{sreg(par[who], e)} = __builtin_amdgcn_exp2f({sreg(par[who], e)});")
        who, e = slot_of(g - 1)
        rb = (e >> 3) & 1
        # (asm volatile: left to the compiler, the whole chain of row-sum adds is sunk to the loop latch - four tiles of
        # probabilities parked in accumulation registers until then)
        if (e & 7) == 0 and (e >> 4) == 0:
            m.add(f'asm volatile("v_mov_b32 %0, %1" : "=v"(lsum[{par[who]}][{rb}]) : "v"({sreg(par[who], e)}));')
        else:
            m.add(f'asm volatile("v_add_f32 %0, %0, %1" : "+v"(lsum[{par[who]}][{rb}]) : "v"({sreg(par[who], e)}));')
        if e & 1:
            m.add(f"pf[{rb}][{e >> 4}][{(e & 7) >> 1}] = pack2<Tag>({sreg(par[who], e - 1)}, {sreg(par[who], e)});")
        if g == 43:   # the late tile's row sums are complete (its last add ran in gap 42)
            for rb in range(2):
                m.add(f'asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(l_run[{rb}]) : "v"(alpha[{P}][{rb}]), "v"(lsum[{P}][{rb}]));')
        # ---- row maxima of the next tile, deferral decision
        if 18 <= g <= 25:
            i = g - 18
            for rb in range(2):
                s_ = f"S[{Pn}][0][{rb}]"
                if i == 0:
                    m.add(f"mxr_[{rb}] = w64_max3({s_}[0], {s_}[1], {s_}[2]);")
                elif i < 7:
                    m.add(f"mxr_[{rb}] = w64_max3(mxr_[{rb}], {s_}[{2 * i + 1}], {s_}[{2 * i + 2}]);")
                else:
                    m.add(f"mxr_[{rb}] = fmaxf(mxr_[{rb}], {s_}[15]);")
        if 34 <= g <= 37:
            for i in (2 * (g - 34), 2 * (g - 34) + 1):
                for rb in range(2):
                    s_ = f"S[{Pn}][1][{rb}]"
                    m.add(f"mxr_[{rb}] = w64_max3(mxr_[{rb}], {s_}[{2 * i}], {s_}[{2 * i + 1}]);")
        if g == 38:
            for rb in range(2):
                m.add(f"mxr_[{rb}] = w64_rowmax_halves(mxr_[{rb}]);")
        if g == 39:
            for rb in range(2):
                m.add(f"mc_[{rb}] = mxr_[{rb}] * sc;")
            m.add("any_ = __any(mc_[0] > m_run[0] + a.defer || mc_[1] > m_run[1] + a.defer);")
        if g == 40:
            for rb in range(2):
                m.add(f"{{ const float mn_ = any_ ? fmaxf(m_run[{rb}], mc_[{rb}]) : m_run[{rb}]; "
                      f"alpha[{Pn}][{rb}] = __builtin_amdgcn_exp2f(m_run[{rb}] - mn_); m_run[{rb}] = mn_; negm[{Pn}][{rb}] = -mn_; }}")
        # ---- the ring: index loads of tile t+5, pieces of tile t+3, the tile's barrier
        if 44 <= g <= 47:
            m.add(f"load_slot({refill}, t + 4, {g - 44});")
        if 48 <= g <= 55:
            m.add(f"dma_piece({issue}, {bufd}, {(g - 48) >> 1}, {'true' if (g - 48) & 1 else 'false'});")
        if g == 59:
            m.add("__builtin_amdgcn_s_waitcnt(0x0078);")
            m.add('asm volatile("s_barrier" ::: "memory");')
        m.add("__builtin_amdgcn_sched_barrier(0);")
    m.add("if (any_) { SP_W64_MFMA_FENCE(); SP_W64_RESCALE_O_0(alpha[%d][0]); SP_W64_RESCALE_O_1(alpha[%d][1]); SP_W64_ACCWRITE_FENCE(); }" % (Pn, Pn))
    m.add("}")
    return m.render()


def _lds_sequence(order):
    """LDS ops of two consecutive iterations in program order: (absolute gap, op); a gap's reads follow its MFMA wait"""
    return [(g - 64, r) for g, r in order] + list(order)


def wait_count_k(order, gap, f):
    seq = _lds_sequence(order)
    # the read that feeds frag f's MFMAs at `gap` of the second iteration
    want_gap = 2 * f - 4 if f >= 2 else (60 - 64 + 2 * f)
    idx = [i for i, (g, r) in enumerate(seq) if g == want_gap and r[0] == "K" and r[1] == f]
    assert len(idx) == 1, (gap, f, idx)
    return sum(1 for i, (g, r) in enumerate(seq) if i > idx[0] and g < gap)


def wait_count_v(order, gap, v):
    seq = _lds_sequence(order)
    want_gap = 25 + 2 * v
    idx = [i for i, (g, r) in enumerate(seq) if g == want_gap and r[0] == "V" and r[1] == v and r[2] == 1]
    assert len(idx) == 1
    return sum(1 for i, (g, r) in enumerate(seq) if i > idx[0] and g < gap)


def generate():
    head = ("// GENERATED by tools/gen_extend_w64.py - do not edit; tests/test_extend_isa.py checks that this file is the\n"
            "// generator's output.  Register-literal parts of extend_w64.hip (see the generator's docstring).\n"
            "// clang-format off\n")
    return head + gen_static() + "".join(steady(b) for b in range(4))


if __name__ == "__main__":
    text = generate()
    if "--check" in sys.argv:
        ok = os.path.exists(OUT) and open(OUT).read() == text
        print("up to date" if ok else "STALE: run tools/gen_extend_w64.py")
        sys.exit(0 if ok else 1)
    open(OUT, "w").write(text)
    print(f"wrote {OUT}: {len(text.splitlines())} lines")
