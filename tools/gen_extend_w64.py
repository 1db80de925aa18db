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
  SP_W64_STEADYM / STEADYD / LEAVE          the same with the ring position as a run-time value (`cb_`): the masked and
                                            the plain form of the last iterations, the way out of the pipeline
  SP_W64_ENTER / ENTERM                     the way in (tile 0 as the next tile of an empty iteration)

The bodies name three macros the including file defines: SP_W64_ENTER_HOOK(g) / SP_W64_LEAVE_HOOK(g) in every gap g of
the ways in and out (the persistent form of the kernel puts the next item's loads there; empty otherwise) and
SP_W64_COLD_WAIT, the vector-memory wait of the run-time-positioned bodies (a full drain, or the persistent form's
counted one).

Usage: python tools/gen_extend_w64.py [--check]   (--check: exit 1 if the committed file differs)
       ... --stamp-gaps 30,31,... / --stamp-masked   diagnostic stamps (-DSP_W64_STAMPS builds): other gaps than
                                                     the default ten segments / in the masked body instead of the
                                                     pipelined ones; write the file, build the variant, regenerate
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
            m.add(f'asm volatile({txt});')
    out.append(m.render())
    m = Macro("SP_W64_SET_Q(QF)")
    for rb in range(2):
        for ks in range(8):
            n = 128 + (rb * 8 + ks) * 4
            txt = " ".join(f'"v_accvgpr_write_b32 a{n + i}, %{i}\\n"' for i in range(4))
            ins = ", ".join(f'"v"(QF[{rb}][{ks}][{i}])' for i in range(4))
            m.add(f'asm volatile({txt} :: {ins});')
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
            m.add(f'asm volatile({" ".join(txt)} : "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_) : "v"(AL));')
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
V0 = 20   # gap of the first V^T read of an iteration (two reads per fragment, 16 fragments, used from gap 32 on)
E0 = 17   # exponentials a tile gets in the iteration BEFORE the one that finishes it (the hand-over state, also in
          # extend_w64.hip as SP_W64_E0): the decision about a tile's maximum is ready around gap 45 of that iteration


def slot_of(n):
    """exponential slot n of an iteration -> (which tile, e): slots 0 .. 63-E0 finish the tile in its late stage
    ('cur', e = n + E0), the last E0 slots start the next one ('nxt'); -1 and 64 are the neighbours' edge slots"""
    if n < 64 - E0:
        return ("cur", n + E0)
    return ("nxt", n - (64 - E0))


def body(name, buf, mask=False, do_cur=True, do_nxt=True, tnext="(t + 1)", dyn=False):
    """One iteration of the tile pipeline as a macro.  The tile in its late stage ('cur': the rest of its exponentials,
    O^T += V^T . P^T) sits in ring buffer `buf`; the next one ('nxt': S^T, row maxima, the decision about its maximum,
    its first E0 exponentials) in buf + 1.  Variants: `mask` applies the causal / ragged mask to the next tile's
    scores (its vector work is placed later and denser: a row block has two or three such tiles); do_nxt only = the way
    INTO the pipeline (tile 0 as the 'next' tile of an empty iteration, no ring work); do_cur only = the way OUT.
    `dyn`: the ring position is a run-time value (`cb_`, the late tile's buffer) instead of part of the macro's name, the
    late tile sits in S[0] whatever its parity (the caller moves it there) and no pieces are issued: the form of the
    last two or three iterations of a row block, one copy of the code instead of four."""
    steady = do_cur and do_nxt
    P, Pn = buf & 1, (buf & 1) ^ 1
    par = {"cur": P, "nxt": Pn}
    on = {"cur": do_cur, "nxt": do_nxt}
    buf1, buf2, bufd = (buf + 1) % 4, (buf + 2) % 4, (buf + 3) % 4
    if dyn:
        assert buf == 0
        buf1, buf2 = 1, 2      # names of the run-time offsets kb1_ / kb2_
    # two index sets: the pieces of tile t+3 go out from the set loaded one iteration ago, the other one is refilled
    # with tile t+4's indices (an iteration is ~3,000 cycles, an index load well under 1,000)
    refill, issue = [("sA", "sB"), ("sB", "sA")][buf & 1]
    # ---- LDS reads by gap: ("K", frag, tile buffer) / ("V", v, half)
    reads = {g: [] for g in range(64)}
    if do_nxt:
        for f in range(3, 16):
            reads[2 * f - 6].append(("K", f, buf1))
        for f in range(3):      # the tile after next: its first fragments, behind the barrier that publishes it
            reads[58 + 2 * f].append(("K", f, buf2))
    if do_cur:
        for v in range(16):
            reads[V0 + 2 * v].append(("V", v, 0))
            reads[V0 + 1 + 2 * v].append(("V", v, 1))
    # program order of LDS ops over one iteration (a gap's wait and MFMA come first, its reads after them)
    order = []
    for g in range(64):
        for r in reads[g]:
            order.append((g, r))

    def kread(f, tb):
        kb, ks = f >> 3, f & 7
        if dyn:   # tb: 1 = the next tile's buffer, 2 = the one after
            return (f'asm volatile("ds_read_b128 %0, %1 offset:{kb * 32 * ROW_B}" : "=v"(kf[{f % 4}]) : '
                    f'"v"(kaddr[{ks}] + kb{tb}_));')
        off = tb * K_TILE + kb * 32 * ROW_B
        return f'asm volatile("ds_read_b128 %0, %1 offset:{off}" : "=v"(kf[{f % 4}]) : "v"(kaddr[{ks}]));'

    # where the next tile's vector work sits
    if mask:
        mask0, max0, mask1, max1, g_half, g_vote, g_mn, g_alpha = range(17, 25), range(25, 33), range(33, 41), range(41, 45), 45, 45, 46, 46
        per_gap1 = 2
    else:
        mask0, max0, mask1, max1, g_half, g_vote, g_mn, g_alpha = (), range(18, 26), (), range(33, 41), 41, 42, 44, 45
        per_gap1 = 1

    m = Macro(name)
    m.add("{ float mxr_[2], mc_[2], mn_[2]; bool any_ = false;")
    if dyn:
        m.add(f"const uint32_t kb1_ = ((cb_ + 1) & 3) * {K_TILE}, kb2_ = ((cb_ + 2) & 3) * {K_TILE}, vb_ = cb_ * {K_TILE};")
    if mask:
        for rb in range(2):
            m.add(f"const int lim{rb}_ = min(kv_len - 1, row_limit[{rb}]) - {tnext} * 64 - 4 * h;   /* visible: key offset <= lim */")
    if do_nxt and not do_cur:   # nobody read the first fragments ahead
        for f in range(3):
            m.add(kread(f, buf1))
    stamp_at = dict(STAMP_AT) if steady and mask == STAMP_MASKED else {}
    for g in range(64):
        m.add(f"/* gap {g} */")
        if g in stamp_at:
            m.add(f"SP_W64_STAMP({stamp_at[g]});")
        # ---- the MFMA (and the counted waits: volatile text between the reads and the MFMAs that use them - tying a
        # wait to the registers makes hipcc put an s_nop behind it)
        if g < 32:
            if do_nxt:
                kb, ks, rb = g >> 4, (g >> 1) & 7, g & 1
                f = kb * 8 + ks
                if rb == 0 and f % 2 == 0:   # one wait per pair of fragments
                    m.add(f'asm volatile("s_waitcnt lgkmcnt({wait_count_k(order, g, f + 1)})");')
                m.add(mfma_s(Pn, kb, rb, ks, f"kf[{f % 4}]"))
        elif do_cur:
            j = g - 32
            s, db, rb = j >> 3, (j >> 1) & 3, j & 1
            v = s * 4 + db
            if rb == 0 and db == 0:   # one wait per k-step: its eight reads
                m.add(f'asm volatile("s_waitcnt lgkmcnt({wait_count_v(order, g, v + 3)})");')
            m.add(f"{{ u32x4 vf_; vf_[0] = vlo[{s & 1}][{db}][0]; vf_[1] = vlo[{s & 1}][{db}][1]; vf_[2] = vhi[{s & 1}][{db}][0]; "
                  f"vf_[3] = vhi[{s & 1}][{db}][1];")
            m.add("  " + mfma_o(rb, db, "vf_", f"pf[{rb}][{s}]") + " }")
        if do_nxt and not do_cur and g == 32:
            # the way in has no O^T MFMAs: nothing separates the last S^T MFMAs from the vector instructions that read
            # their results (the MFMA's passes + 3 wait states; the compiler cannot see that asm text is an MFMA)
            m.add("SP_W64_MFMA_FENCE();")
        # ---- LDS reads
        for r in reads[g]:
            if r[0] == "K":
                m.add(kread(r[1], r[2]))
            else:
                _, v, half = r
                s, db = v >> 2, v & 3
                off = (0 if dyn else buf * K_TILE) + s * 16 * ROW_B + half * 8 * ROW_B
                dst = f"vhi[{s & 1}][{db}]" if half else f"vlo[{s & 1}][{db}]"
                src = f"va[{db}] + vb_" if dyn else f"va[{db}]"
                m.add(f'asm volatile("ds_read_b64_tr_b16 %0, %1 offset:{off}" : "=v"({dst}) : "v"({src}));')
        # ---- the next tile: mask, row maxima, deferral decision (ahead of the exponent pipeline in the gap's text: the
        # masked form decides in the very gap whose fma is the first to use the new maximum)
        if do_nxt:
            for kbm, gaps in ((0, mask0), (1, mask1)):
                if g in gaps:
                    i0 = (g - gaps[0]) * 4      # four registers of both row blocks per gap
                    for r in range(i0, i0 + 4):
                        koff = kbm * 32 + (r & 3) + 8 * (r >> 2)
                        for rb in range(2):
                            s_ = f"S[{Pn}][{kbm}][{rb}][{r}]"
                            m.add(f"{s_} = {koff} <= lim{rb}_ ? {s_} : -INFINITY;")
            if g in max0:
                i = g - max0[0]
                for rb in range(2):
                    s_ = f"S[{Pn}][0][{rb}]"
                    if i == 0:
                        m.add(f"mxr_[{rb}] = fmaxf(fmaxf({s_}[0], {s_}[1]), {s_}[2]);")
                    elif i < 7:
                        m.add(f"mxr_[{rb}] = fmaxf(fmaxf(mxr_[{rb}], {s_}[{2 * i + 1}]), {s_}[{2 * i + 2}]);")
                    else:
                        m.add(f"mxr_[{rb}] = fmaxf(mxr_[{rb}], {s_}[15]);")
            if g in max1:      # a dependent chain per row block; nothing else waits for it yet
                for i in range((g - max1[0]) * per_gap1, (g - max1[0] + 1) * per_gap1):
                    for rb in range(2):
                        s_ = f"S[{Pn}][1][{rb}]"
                        m.add(f"mxr_[{rb}] = fmaxf(fmaxf(mxr_[{rb}], {s_}[{2 * i}]), {s_}[{2 * i + 1}]);")
            if g == g_half:
                for rb in range(2):
                    m.add(f"mxr_[{rb}] = w64_rowmax_halves(mxr_[{rb}]);")
            if g == g_vote:
                for rb in range(2):
                    m.add(f"mc_[{rb}] = mxr_[{rb}] * sc;")
                m.add("any_ = __any(mc_[0] > m_run[0] + a.defer || mc_[1] > m_run[1] + a.defer);")
            if g == g_mn:      # (behind the vote: the scalar result has to cross to the vector side)
                for rb in range(2):
                    m.add(f"mn_[{rb}] = any_ ? fmaxf(m_run[{rb}], mc_[{rb}]) : m_run[{rb}];")
            if g == g_alpha:
                for rb in range(2):
                    m.add(f"alpha[{Pn}][{rb}] = __builtin_amdgcn_exp2f(m_run[{rb}] - mn_[{rb}]); m_run[{rb}] = mn_[{rb}]; negm[{Pn}][{rb}] = -mn_[{rb}];")
        # ---- exponential pipeline: fma of slot g+1, exp of slot g, add / pack of slot g-1
        who, e = ("nxt", E0) if g == 63 else slot_of(g + 1)
        if on[who]:
            m.add(f"{sreg(par[who], e)} = __builtin_fmaf({sreg(par[who], e)}, sc, negm[{par[who]}][{(e >> 3) & 1}]);")
        who, e = slot_of(g)
        if on[who]:
            m.add(f"{sreg(par[who], e)} = __builtin_amdgcn_exp2f({sreg(par[who], e)});")
        who, e = slot_of(g - 1)
        rb = (e >> 3) & 1
        if on[who]:
            # (asm volatile: left to the compiler, the whole chain of row-sum adds is sunk to the loop latch - four tiles
            # of probabilities parked in accumulation registers until then)
            # The asm form relies on the gap's MFMA (volatile text ahead of it) to separate it from the v_exp that made
            # its operand - a transcendental's result needs one instruction before a vector instruction reads it, and the
            # compiler does not look into asm text.  Gaps without an MFMA (the ways in and out) use plain C++.
            has_mfma = (g < 32 and do_nxt) or (g >= 32 and do_cur)
            first = (e & 7) == 0 and (e >> 4) == 0
            if not has_mfma:
                m.add(f"lsum[{par[who]}][{rb}] {'=' if first else '+='} {sreg(par[who], e)};")
            elif first:
                m.add(f'asm volatile("v_mov_b32 %0, %1" : "=v"(lsum[{par[who]}][{rb}]) : "v"({sreg(par[who], e)}));')
            else:
                m.add(f'asm volatile("v_add_f32 %0, %0, %1" : "+v"(lsum[{par[who]}][{rb}]) : "v"({sreg(par[who], e)}));')
            if e & 1:
                m.add(f"pf[{rb}][{e >> 4}][{(e & 7) >> 1}] = pack2<Tag>({sreg(par[who], e - 1)}, {sreg(par[who], e)});")
        if do_cur and g == 64 - E0 + 1:   # the late tile's row sums are complete (its last add ran one gap earlier)
            for rb in range(2):
                m.add(f'asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(l_run[{rb}]) : "v"(alpha[{P}][{rb}]), "v"(lsum[{P}][{rb}]));')
        # ---- the ring: pieces of tile t+3 early in the iteration (its buffer was retired by the last barrier; the gaps of
        # phase A carry the fewest fillers), then the index loads of tile t+4, the tile's barrier near the end
        if do_nxt and not do_cur and g == 57:
            # the way in starts as soon as Q and tile 0's K have landed; the rest of the prologue's pieces (tile 0's V, tile
            # 1) must have by now: everything but the 12 youngest operations (tile 2's pieces, tile 3's indices)
            m.add("__builtin_amdgcn_s_waitcnt(0x007C);")
            m.add('asm volatile("s_barrier" ::: "memory");')
        if steady and dyn and g == 57:
            # no pieces in this form (tile t+3 lies past the row block's last tile): drain and publish tile t+2
            if STAMP_WAIT is not None and stamp_at:
                m.add(f"SP_W64_STAMP({STAMP_WAIT});")
            m.add("SP_W64_COLD_WAIT")
            m.add('asm volatile("s_barrier" ::: "memory");')
        if steady and not dyn:
            if 2 <= g <= 16 and g % 2 == 0:
                k = (g - 2) // 2
                m.add(f"dma_piece({issue}, {bufd}, {k >> 1}, {'true' if k & 1 else 'false'});")
            if 26 <= g <= 29:
                m.add(f"load_slot({refill}, t + 4, {g - 26});")
            if g == 57:
                if STAMP_WAIT is not None and stamp_at:
                    m.add(f"SP_W64_STAMP({STAMP_WAIT});")
                # everything but the 16 youngest vector-memory operations: the index loads of tile t+3 (last iteration),
                # this iteration's 8 pieces and 4 index loads; i.e. tile t+2's pieces have landed.  lgkmcnt(0).
                m.add("__builtin_amdgcn_s_waitcnt(0x4070);")
                m.add('asm volatile("s_barrier" ::: "memory");')
        # hooks of the persistent form of the kernel (empty macros elsewhere): work for the NEXT plan item spread over the
        # gaps of the ways in and out - between two bodies it would sit in front of the next body's first wait
        if do_nxt and not do_cur:
            m.add(f"SP_W64_ENTER_HOOK({g})")
        if do_cur and not do_nxt:
            m.add(f"SP_W64_LEAVE_HOOK({g})")
        m.add("__builtin_amdgcn_sched_barrier(0);")
    if stamp_at:
        m.add(f"SP_W64_STAMP({STAMP_END}); SP_W64_STAMP_ACC();")
    if steady:   # (on the way in O^T is still zero: nothing to rescale)
        # (rare - the deferred maximum moves in a row block's first tiles: out of line, the loop's code stays dense)
        m.add("if (__builtin_expect(any_, 0)) { SP_W64_MFMA_FENCE(); SP_W64_RESCALE_O_0(alpha[%d][0]); SP_W64_RESCALE_O_1(alpha[%d][1]); SP_W64_ACCWRITE_FENCE(); }" % (Pn, Pn))
    m.add("}")
    return m.render()


def bodies():
    out = []
    for b in range(4):
        out.append(body(f"SP_W64_STEADY_{b}", b))
    out.append(body("SP_W64_STEADYM", 0, mask=True, dyn=True))
    out.append(body("SP_W64_STEADYD", 0, dyn=True))
    out.append(body("SP_W64_LEAVE", 0, do_nxt=False, dyn=True))
    # the way in: tile 0 is the 'next' tile of an empty iteration in ring position 3
    out.append(body("SP_W64_ENTER", 3, do_cur=False, tnext="0"))
    out.append(body("SP_W64_ENTERM", 3, mask=True, do_cur=False, tnext="0"))
    return "".join(out)


def _lds_sequence(order):
    """LDS ops of two consecutive iterations in program order: (absolute gap, op); a gap's reads follow its MFMA wait"""
    return [(g - 64, r) for g, r in order] + list(order)


def wait_count_k(order, gap, f):
    seq = _lds_sequence(order)
    # the read that feeds frag f's MFMAs in the second iteration
    want_gap = 2 * f - 6 if f >= 3 else (58 - 64 + 2 * f)
    idx = [i for i, (g, r) in enumerate(seq) if g == want_gap and r[0] == "K" and r[1] == f]
    assert len(idx) == 1, (gap, f, idx)
    return sum(1 for i, (g, r) in enumerate(seq) if i > idx[0] and g < gap)


def wait_count_v(order, gap, v):
    seq = _lds_sequence(order)
    want_gap = V0 + 1 + 2 * v
    idx = [i for i, (g, r) in enumerate(seq) if g == want_gap and r[0] == "V" and r[1] == v and r[2] == 1]
    assert len(idx) == 1
    return sum(1 for i, (g, r) in enumerate(seq) if i > idx[0] and g < gap)


# diagnostic stamps (-DSP_W64_STAMPS builds only): stamp index by gap; the default splits an iteration into ten segments.
# `--stamp-gaps 30,31,...` (at most 22 gaps, ascending) replaces them, e.g. one stamp per gap of a stretch.
STAMP_AT = {0: 0, 8: 1, 16: 2, 24: 3, 32: 4, 40: 5, 48: 6, 56: 7, 60: 9}
STAMP_WAIT = 8
STAMP_END = 10
STAMP_MASKED = False      # --stamp-masked: the stamps go into the masked run-time-positioned body instead


def generate():
    global STAMP_AT, STAMP_WAIT, STAMP_END, STAMP_MASKED
    STAMP_MASKED = "--stamp-masked" in sys.argv
    for i, arg in enumerate(sys.argv):
        if arg == "--stamp-gaps":
            gaps = [int(x) for x in sys.argv[i + 1].split(",")]
            assert len(gaps) <= 22 and gaps == sorted(gaps)
            STAMP_AT = {g: k for k, g in enumerate(gaps)}
            STAMP_WAIT = None
            STAMP_END = len(gaps)
    head = ("// GENERATED by tools/gen_extend_w64.py - do not edit; tests/test_extend_isa.py checks that this file is the\n"
            "// generator's output.  Register-literal parts of extend_w64.hip (see the generator's docstring).\n"
            "// clang-format off\n")
    return head + f"#define SP_W64_NSTAMP {STAMP_END + 1}\n#define SP_W64_E0 {E0}\n" + gen_static() + bodies()


if __name__ == "__main__":
    text = generate()
    if "--stdout" in sys.argv:
        sys.stdout.write(text)
        sys.exit(0)
    if "--check" in sys.argv:
        ok = os.path.exists(OUT) and open(OUT).read() == text
        print("up to date" if ok else "STALE: run tools/gen_extend_w64.py")
        sys.exit(0 if ok else 1)
    open(OUT, "w").write(text)
    print(f"wrote {OUT}: {len(text.splitlines())} lines")
