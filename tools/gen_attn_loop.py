#!/usr/bin/env python3
"""Generate the hand-scheduled KV loop of the 8-wave region attention kernel (regione_amd/csrc/attn_loop_asm.inc).

    python tools/gen_attn_loop.py            # rewrites the .inc (committed; the build does not run this script)

What the compiler-scheduled kernel (attention_kernel, attn.hip) does per KV tile and wave: 16 MFMAs (S^T = K Q^T), then
~150 VALU instructions of softmax with the matrix pipe idle, then 16 MFMAs (O^T += V^T P^T).  Both waves of a SIMD pass the
same barrier every tile, so their softmax phases coincide: a SIMD spends ~3780 cycles on 2048 cycles of MFMA work
(DESIGN.md section 4.7).  hipcc cannot be made to software-pipeline this (round 1 tried).  Here the loop is ONE asm
statement in which each wave overlaps its OWN phases:

    X(t):  MFMA  S(t+1) = K(t+1) Q^T            ||  VALU  P(t) = exp2(S(t) * c - m),  P -> bf16 (pf)
    Y(t):  MFMA  O += V^T(t) P(t)^T            ||  VALU  row sums of P(t),  max of S(t+1),  defer-max decision

Two S register sets alternate (tile parity), so the loop body exists in two parities.  K / V^T fragments stream through an
8-slot window (32 registers), each fragment requested 4 MFMAs before its use, with counted lgkmcnt waits.  K / V^T tiles
arrive by LDS-DMA into a ring of FIVE 32 KiB stages (all 160 KiB): when a wave passes the tile barrier of body t every
wave's pieces of tiles <= t+2 have landed, tile t+3 may be in flight and tile t+4 is issued right behind the barrier
(two tiles of lead; the fragment prefetch may run across the barrier).  Past the last tile the DMA re-fetches the last tile
(clamped offset), which keeps every `vmcnt` count uniform.  Online softmax with deferred max exactly like the C++ kernel
(running max raised - and O, l rescaled - only when some row's tile max exceeds it by > 2^8); the rescale sits at the head of
X(t+1), after PV(t) has drained.

Registers (per wave; 2 waves per SIMD -> 256 in total):
    a[0:63]    O^T accumulator (4 d-blocks x 16)          a[64:95]  Q fragments (B operand of S^T), 8 k-steps x 4
    v[32:63]   S set 0 (s0 = kv rows 0..31, s1 = 32..63)   v[64:95]  S set 1
    v[96:111]  pf: P packed to bf16 (B operand of PV)      v[112:143] fragment window, 8 slots x 4
    v[144:159] temporaries                                 v[0:31]   operands (compiler-allocated)
Named operands: see attn.hip (attention_asm_kernel).
"""
import os

S_BASE = (32, 64)
PF, FR, TMP = 96, 112, 144
T_NEGM, T_MX, T_MXB, T_KA, T_VA, T_A, T_B, T_ALPHA = (TMP + i for i in range(8))
ACC = [TMP + 8 + i for i in range(4)]           # row-sum accumulators
T_R = [TMP + 12 + i for i in range(4)]          # rescale temporaries
K_TILE, STAGE, NSTAGE = 16384, 32768, 5
LDS_END = STAGE * NSTAGE
CFG = dict(D=4, novalu=False, noread=False, prio=False, pkfma=False, pkadd=False, dot2sum=False, dephase=False, dmaspread=False,
           staticmax=False)      # generator knobs (main() emits several variants)


def v(n, w=1):
    return f"v{n}" if w == 1 else f"v[{n}:{n + w - 1}]"


def a(n, w=1):
    return f"a{n}" if w == 1 else f"a[{n}:{n + w - 1}]"


def slot(i):
    return v(FR + 4 * (i % 8), 4)


class Stream:
    """Instruction list with LDS-read bookkeeping for counted lgkmcnt waits."""

    def __init__(self):
        self.ins = []
        self.pending = []            # fragment tags of ds_reads issued and not yet known complete (in issue order)

    def emit(self, s):
        self.ins.append(s)

    def read(self, tag, dst, addr, off):
        if CFG["noread"]:
            return
        self.ins.append(f"ds_read_b128 {dst}, {addr} offset:{off}")
        self.pending.append(tag)

    def need(self, tag):
        """make sure the read `tag` has completed: LDS reads return in order"""
        if tag in self.pending:
            i = self.pending.index(tag)
            n = len(self.pending) - 1 - i
            self.ins.append(f"s_waitcnt lgkmcnt({n})")
            self.pending = self.pending[i + 1:]


def k_addr(st, ks):
    st.emit(f"v_add_u32 {v(T_KA)}, %[stg_k], %[krel{ks}]")


def v_addr(st, kb4):
    st.emit(f"v_add_u32 {v(T_VA)}, %[stg_v], %[vrel{kb4}]")


def k_read(st, f):                 # K fragment f = 2*ks + b  (b = kv half of the tile)
    ks, b = divmod(f, 2)
    if b == 0:
        k_addr(st, ks)
    st.read(("K", f), slot(f), v(T_KA), b * 8192)


def v_read(st, g):                 # V^T fragment g = 4*kb4 + db
    kb4, db = divmod(g, 4)
    if db == 0:
        v_addr(st, kb4)
    st.read(("V", g), slot(g), v(T_VA), K_TILE + db * 4096)


def qk_mfma(st, f, q):
    ks, b = divmod(f, 2)
    d = v(S_BASE[q] + 16 * b, 16)
    c = "0" if ks == 0 else d
    st.need(("K", f))
    st.emit(f"v_mfma_f32_32x32x16_bf16 {d}, {slot(f)}, {a(64 + 4 * ks, 4)}, {c}")


def pv_mfma(st, g):
    kb4, db = divmod(g, 4)
    st.need(("V", g))
    st.emit(f"v_mfma_f32_32x32x16_bf16 {a(16 * db, 16)}, {slot(g)}, {v(PF + 4 * kb4, 4)}, {a(16 * db, 16)}")


def exp_ops(p):
    """VALU of X(t) on S set p, in place: P = exp2(S * c - m); pf[k] = bf16x2(P[2k], P[2k+1]).  Ordered so that an op never
    follows its producer closely.  ATTN_PKFMA=1 / ATTN_PKADD=1 (measurement builds): scale-and-shift / row sums on the
    packed-fp32 instructions (two elements each; 35 VALU instructions fewer per tile) - measured SLOWER on MI355X (full step
    1175 -> 1136 / 1148 / 1113 TFLOP/s with pk_fma / pk_add / both, same box), so the shipped loop keeps the scalar forms."""
    b = S_BASE[p]
    if not CFG["pkfma"]:
        return exp_ops_scalar(p)
    assert T_NEGM % 2 == 0 and b % 2 == 0
    fma = lambda i: f"v_pk_fma_f32 {v(b + 2 * i, 2)}, {v(b + 2 * i, 2)}, %[sl2e2], {v(T_NEGM, 2)} op_sel_hi:[1,1,0]"
    exp = lambda i: f"v_exp_f32 {v(b + i)}, {v(b + i)}"
    cvt = lambda k: f"v_cvt_pk_bf16_f32 {v(PF + k)}, {v(b + 2 * k)}, {v(b + 2 * k + 1)}"
    ops = [fma(i) for i in range(4)]
    for i in range(32 + 12):
        if i < 32:
            ops.append(exp(i))
        if i % 2 == 1 and (i + 8) // 2 < 16:
            ops.append(fma((i + 8) // 2))              # elements i+7, i+8: needed by exp(i+7) six or more ops later
        j = i - 10
        if 0 <= j < 32 and j % 2 == 1:
            ops.append(cvt(j // 2))
    assert sum(o.startswith("v_cvt") for o in ops) == 16 and sum(o.startswith("v_pk_fma") for o in ops) == 16
    # every element is scaled before its exp
    seen = set()
    for o in ops:
        if o.startswith("v_pk_fma"):
            r = int(o.split("v[")[1].split(":")[0])
            seen |= {r, r + 1}
        elif o.startswith("v_exp"):
            assert int(o.split()[1].strip("v,")) in seen, o
    return ops


def rowsum_ops(p):
    """l_run += sum of the 32 P values of this lane: two packed accumulators (4 running sums), 17 instructions"""
    b = S_BASE[p]
    if CFG["dot2sum"]:
        # ATTN_DOT2SUM=1 (measurement build): row sum of the bf16-ROUNDED P (the values the PV product uses) with
        # v_dot2_f32_bf16 against (1, 1): 16 + 4 instructions instead of 36
        ops = [f"v_dot2_f32_bf16 {v(ACC[i % 4])}, {v(PF + i)}, %[ones2], " + ("0" if i < 4 else v(ACC[i % 4])) for i in range(16)]
        ops += [f"v_add_f32 {v(ACC[0])}, {v(ACC[0])}, {v(ACC[1])}", f"v_add_f32 {v(ACC[2])}, {v(ACC[2])}, {v(ACC[3])}",
                f"v_add_f32 {v(ACC[0])}, {v(ACC[0])}, {v(ACC[2])}", f"v_add_f32 %[l_run], %[l_run], {v(ACC[0])}"]
        return ops
    if not CFG["pkadd"]:
        return rowsum_ops_scalar(p)
    assert ACC[0] % 2 == 0
    acc = [v(ACC[0], 2), v(ACC[2], 2)]
    ops = [f"v_pk_add_f32 {acc[0]}, {v(b, 2)}, {v(b + 2, 2)}", f"v_pk_add_f32 {acc[1]}, {v(b + 4, 2)}, {v(b + 6, 2)}"]
    ops += [f"v_pk_add_f32 {acc[(i // 2) % 2]}, {acc[(i // 2) % 2]}, {v(b + i, 2)}" for i in range(8, 32, 2)]
    ops += [f"v_pk_add_f32 {acc[0]}, {acc[0]}, {acc[1]}", f"v_add_f32 {v(ACC[0])}, {v(ACC[0])}, {v(ACC[1])}",
            f"v_add_f32 %[l_run], %[l_run], {v(ACC[0])}"]
    return ops


def exp_ops_scalar(p):
    b = S_BASE[p]
    fma = lambda i: f"v_fma_f32 {v(b + i)}, {v(b + i)}, %[sl2e], {v(T_NEGM)}"
    exp = lambda i: f"v_exp_f32 {v(b + i)}, {v(b + i)}"
    cvt = lambda k: f"v_cvt_pk_bf16_f32 {v(PF + k)}, {v(b + 2 * k)}, {v(b + 2 * k + 1)}"
    ops = [fma(i) for i in range(8)]
    for i in range(32 + 12):
        if i < 32:
            ops.append(exp(i))
        if i + 8 < 32:
            ops.append(fma(i + 8))
        j = i - 10
        if 0 <= j < 32 and j % 2 == 1:
            ops.append(cvt(j // 2))
    return ops


def rowsum_ops_scalar(p):
    b = S_BASE[p]
    ops = [f"v_mov_b32 {v(ACC[i])}, {v(b + i)}" for i in range(4)]
    ops += [f"v_add_f32 {v(ACC[i % 4])}, {v(ACC[i % 4])}, {v(b + i)}" for i in range(4, 32)]
    ops += [f"v_add_f32 {v(ACC[0])}, {v(ACC[0])}, {v(ACC[1])}", f"v_add_f32 {v(ACC[2])}, {v(ACC[2])}, {v(ACC[3])}",
            f"v_add_f32 {v(ACC[0])}, {v(ACC[0])}, {v(ACC[2])}", f"v_add_f32 %[l_run], %[l_run], {v(ACC[0])}"]
    return ops


def max_ops(q):
    """tile max of S set q (two chains), across the two half-waves, scaled; vcc = rows whose max exceeds m_run + 8"""
    b = S_BASE[q]
    ops = [f"v_max_f32 {v(T_MX)}, {v(b)}, {v(b + 1)}", f"v_max_f32 {v(T_MXB)}, {v(b + 16)}, {v(b + 17)}"]
    for i in range(1, 8):
        ops.append(f"v_max3_f32 {v(T_MX)}, {v(T_MX)}, {v(b + 2 * i)}, {v(b + 2 * i + 1)}")
        ops.append(f"v_max3_f32 {v(T_MXB)}, {v(T_MXB)}, {v(b + 16 + 2 * i)}, {v(b + 17 + 2 * i)}")
    ops += [f"v_max_f32 {v(T_MX)}, {v(T_MX)}, {v(T_MXB)}",
            f"v_mov_b32 {v(T_A)}, {v(T_MX)}", f"v_mov_b32 {v(T_B)}, {v(T_MX)}", "s_nop 1",
            f"v_permlane32_swap_b32 {v(T_A)}, {v(T_B)}",               # lanes 32-63 of A <-> lanes 0-31 of B
            f"v_max_f32 {v(T_MX)}, {v(T_A)}, {v(T_B)}",
            f"v_mul_f32 {v(T_MX)}, %[sl2e], {v(T_MX)}",
            f"v_add_f32 {v(T_A)}, 0x41000000, %[m_run]",                  # m_run + 8.0 (DEFER_THR)
            f"v_cmp_gt_f32 vcc, {v(T_MX)}, {v(T_A)}"]
    return ops


def rescale_block(st, tag):
    """head of X: if any row's new tile max exceeds the running max by more than 2^8 (vcc from the previous Y), raise the
    running max and rescale l and O.  PV of the previous tile has issued; its results need ~20 wait states."""
    st.emit(f"s_cbranch_vccz 9{tag}f")
    st.emit("s_nop 15")
    st.emit("s_nop 15")
    st.emit(f"v_max_f32 {v(T_A)}, %[m_run], {v(T_MX)}")                 # m_new
    st.emit(f"v_sub_f32 {v(T_ALPHA)}, %[m_run], {v(T_A)}")
    st.emit(f"v_exp_f32 {v(T_ALPHA)}, {v(T_ALPHA)}")                    # alpha = 2^(m_old - m_new)
    st.emit(f"v_mov_b32 %[m_run], {v(T_A)}")
    st.emit(f"v_sub_f32 {v(T_NEGM)}, 0, {v(T_A)}")
    st.emit(f"v_mul_f32 %[l_run], %[l_run], {v(T_ALPHA)}")
    for i in range(0, 64, 4):
        for k in range(4):
            st.emit(f"v_accvgpr_read_b32 {v(T_R[k])}, {a(i + k)}")
        for k in range(4):
            st.emit(f"v_mul_f32 {v(T_R[k])}, {v(T_R[k])}, {v(T_ALPHA)}")
        for k in range(4):
            st.emit(f"v_accvgpr_write_b32 {a(i + k)}, {v(T_R[k])}")
    st.emit(f"9{tag}:")


def advance(st, reg):
    st.emit(f"s_add_u32 %[{reg}], %[{reg}], {STAGE}")
    st.emit(f"s_cmp_ge_u32 %[{reg}], {LDS_END}")
    st.emit(f"s_cselect_b32 %[stmp], {LDS_END}, 0")
    st.emit(f"s_sub_u32 %[{reg}], %[{reg}], %[stmp]")


def dma_issue(st):
    """4 pieces of the next tile (clamped to the last one) into stage stg_d; advance the tile offsets and stg_d"""
    st.emit("s_min_u32 %[stmp], %[tk], %[tk_last]")
    st.emit("s_min_u32 %[stmp2], %[tv], %[tv_last]")
    st.emit("s_add_u32 %[sdst], %[stg_d], %[wdst]")
    for p in range(2):
        st.emit(f"s_add_u32 m0, %[sdst], {p * 1024}")
        st.emit("s_nop 0")
        st.emit(f"buffer_load_dwordx4 %[dk{p}], %[rk], %[stmp] offen lds")
    for p in range(2):
        st.emit(f"s_add_u32 m0, %[sdst], {K_TILE + p * 1024}")
        st.emit("s_nop 0")
        st.emit(f"buffer_load_dwordx4 %[dv{p}], %[rv], %[stmp2] offen lds")
    st.emit("s_add_u32 %[tk], %[tk], %[kadv]")
    st.emit("s_add_u32 %[tv], %[tv], 128")
    advance(st, "stg_d")


def dma_split():
    """dma_issue cut into a prologue (scalar set-up), the four pieces (callables) and an epilogue (offset / stage advance): the
    pieces are issued one at a time behind QK MFMAs 0, 2, 4, 6 of phase X (ATTN_DMA_AT) instead of back to back behind the tile
    barrier, where nothing is queued on the matrix pipe yet (an LDS-DMA piece costs the issuing wave ~60 cycles among MFMAs already
    queued, more otherwise).  Round 3, same box: full step 1202 -> 1212 TFLOP/s, region 25 % 965 -> 982, v1p2 2048^2 region 1203 -> 1220;
    placements 1,5,9,13 / 3,7,11,15 / 9,11,13,15 within 0.5 % of it.  ATTN_DMA_SPREAD=0 regenerates the round-2 schedule."""
    def pro(st):
        st.emit("s_min_u32 %[stmp], %[tk], %[tk_last]")
        st.emit("s_min_u32 %[stmp2], %[tv], %[tv_last]")
        st.emit("s_add_u32 %[sdst], %[stg_d], %[wdst]")

    def piece(kind, p):
        def f(st):
            st.emit(f"s_add_u32 m0, %[sdst], {(K_TILE if kind == 'v' else 0) + p * 1024}")
            st.emit("s_nop 0")
            st.emit(f"buffer_load_dwordx4 %[d{kind}{p}], %[r{kind}], %[{'stmp2' if kind == 'v' else 'stmp'}] offen lds")
        return f

    def epi(st):
        st.emit("s_add_u32 %[tk], %[tk], %[kadv]")
        st.emit("s_add_u32 %[tv], %[tv], 128")
        advance(st, "stg_d")
    return pro, [piece("k", 0), piece("k", 1), piece("v", 0), piece("v", 1)], epi


def interleave(st, mfmas, fillers, per_gap):
    """emit the MFMA callbacks with `per_gap[i]` fillers behind MFMA i (fillers: strings or callables(st))"""
    fi = 0
    for i, m in enumerate(mfmas):
        m(st)
        for _ in range(per_gap[i] if i < len(per_gap) else 0):
            if fi < len(fillers):
                f = fillers[fi]
                fi += 1
                f(st) if callable(f) else st.emit(f)
    while fi < len(fillers):
        f = fillers[fi]
        fi += 1
        f(st) if callable(f) else st.emit(f)


def spread(n_fill, n_gaps, first=0):
    """distribute n_fill fillers over n_gaps gaps as evenly as possible, none before gap `first`"""
    gaps = [0] * n_gaps
    live = n_gaps - first
    for k in range(n_fill):
        gaps[first + (k * live) // max(n_fill, 1)] += 1
    return gaps


def phase_x(st, p, full, dma=None):
    """X(t), S(t) in set p: exp of tile t; with `full` also the QK MFMAs of tile t+1 into set 1-p.
    Entry: K fragments 0..D-1 of tile t+1 requested (slots 0..D-1).  Exit: V fragments 0..D-1 of tile t requested.
    Fragment i of a phase lives in slot i % 8; it is requested D MFMAs before its use: behind MFMA i-D-1... i.e. read(i + D)
    follows MFMA i - 1 / precedes MFMA i; the first D fragments of the NEXT phase are requested behind this phase's
    MFMA j + 8 (the last user of slot j)."""
    D = CFG["D"]
    q = 1 - p
    vops = [] if CFG["novalu"] else exp_ops(p)
    if not full:
        for o in vops:
            st.emit(o)
        for g in range(D):
            v_read(st, g)
        return
    mf = []
    at = [int(x) for x in os.environ.get("ATTN_DMA_AT", "0,2,4,6").split(",")]
    for f in range(16):
        def m(st, f=f):
            if f + D < 16:
                k_read(st, f + D)
            qk_mfma(st, f, q)
            if f >= 8 and f - 8 < D:                       # slot f-8 is free now: V prefetch for Y(t)
                v_read(st, f - 8)
            if dma is not None and f in at:                # one LDS-DMA piece of tile t+4 behind this MFMA
                pieces, epi = dma
                pieces[at.index(f)](st)
                if f == at[-1]:
                    epi(st)
        mf.append(m)
    gaps = spread(len(vops), 16)
    interleave(st, mf, vops, gaps)


def phase_y(st, p, full, prefetch_k):
    """Y(t): PV MFMAs of tile t (P in set p / pf); row sums of P(t); with `full` the max + defer decision of S(t+1) (set 1-p);
    with `prefetch_k` the first D K fragments of tile t+2 for the next X.  Entry: V fragments 0..D-1 requested."""
    D = CFG["D"]
    q = 1 - p
    vops = [] if CFG["novalu"] else rowsum_ops(p)
    mops = max_ops(q) if (full and not CFG["staticmax"]) else []
    if CFG["novalu"] and mops:
        mops = mops[-1:]                                    # keep vcc defined
    mf = []
    for g in range(16):
        def m(st, g=g):
            if g + D < 16:
                v_read(st, g + D)
            pv_mfma(st, g)
            if prefetch_k and g >= 8 and g - 8 < D:         # slot g-8 is free now: K prefetch for X(t+1)
                k_read(st, g - 8)
        mf.append(m)
    # the max reads S(t+1), written by the last QK MFMA of X(t): keep it behind the 4th PV MFMA
    fill = vops[:8] + mops + vops[8:] if mops else vops
    gaps = spread(len(fill), 16, first=3 if mops else 0)
    interleave(st, mf, fill, gaps)


def body(st, p, full, tag, role="A"):
    """One KV tile.  Role A: tile barrier + DMA issue at the HEAD of the body (before X); role B: between X and Y.  Both roles
    execute the same phase sequence and one barrier per tile, so when all waves meet at barrier t the role-A waves are about to
    start X(t) and the role-B waves have just finished it: behind the barrier A runs X (80 VALU instructions, 2/3 of them
    quarter-rate exp) while B runs Y (61 cheap ones), then the other way round.  Two waves of one SIMD with different roles
    therefore never want the VALU for their exp phases at the same time.  ATTN_DEPHASE=1 (measurement build; the kernel picks the
    roles, waves 4..7 or the odd waves in role B): correct, and within +-0.5 % of the single-role loop on
    every shape (MI355X) - the waves of a SIMD evidently do not stay phase-locked behind the barrier anyway - so the shipped loop
    has one role.  Ring safety is unchanged: at barrier t every wave has
    finished Y(t-1) (the stage tile t+4 overwrites), and role B's early X(t) reads K(t+1), which landed by barrier t-1."""
    if not CFG["staticmax"]:
        rescale_block(st, tag)

    spread_dma = CFG.get("dmaspread") and role == "A"
    dma = None

    def tile_barrier():
        nonlocal dma
        st.emit("s_waitcnt vmcnt(4)")                      # tile t+2 has landed (t+3 may be in flight)
        st.emit("s_barrier")
        if spread_dma:
            pro, pieces, epi = dma_split()
            pro(st)
            dma = (pieces, epi)                            # issued one by one inside phase X
        else:
            dma_issue(st)                                  # tile t+4 -> the stage of tile t-1
    if full and role == "A":
        tile_barrier()
    phase_x(st, p, full, dma)
    if full:
        advance(st, "stg_k")                               # next X reads K of tile t+2
    if full and role == "B":
        tile_barrier()
    phase_y(st, p, full, prefetch_k=full)
    advance(st, "stg_v")
    # a trailing full body's K prefetch reads (t+2) are harmless when tile t+2 does not exist: the clamped DMA re-fetched
    # the last tile into that stage


def prologue(st):
    # Q fragments -> fragment window (plain loads), then the first four tiles
    for ks in range(8):
        st.emit(f"global_load_dwordx4 {slot(ks)}, %[qptr], off offset:{ks * 32}")
    for _ in range(4):
        dma_issue(st)
    for n in range(64):
        st.emit(f"v_accvgpr_write_b32 {a(n)}, 0")
    st.emit("s_waitcnt vmcnt(16)")                         # Q landed (the 16 DMA pieces may be in flight)
    for ks in range(8):
        for k in range(4):
            st.emit(f"v_accvgpr_write_b32 {a(64 + 4 * ks + k)}, {v(FR + 4 * ks + k)}")
    st.emit(f"v_sub_f32 {v(T_NEGM)}, 0, %[m_run]")
    st.emit("s_waitcnt vmcnt(8)")                          # tiles 0, 1 landed
    st.emit("s_barrier")
    # S(0) = K(0) Q^T into set 0 (stage 0: stg_v), not overlapped
    st.emit("s_mov_b32 %[stmp], %[stg_k]")
    st.emit("s_mov_b32 %[stg_k], %[stg_v]")
    D = CFG["D"]
    for f in range(D):
        k_read(st, f)
    for f in range(16):
        if f + D < 16:
            k_read(st, f + D)
        qk_mfma(st, f, 0)
    st.emit("s_mov_b32 %[stg_k], %[stmp]")
    st.emit("s_nop 15")
    st.emit("s_nop 15")
    if not CFG["staticmax"]:
        for o in max_ops(0):
            st.emit(o)
    for f in range(D):                                      # K prefetch of tile 1 for X(0)
        k_read(st, f)


def emit_loop(st, entry, role, L):
    """the tile loop for one role; L = label base (numeric local labels L+1 .. L+4, rescale labels 9<tag>)"""
    t = lambda k: str(L + k)
    st.pending = list(entry)
    st.emit("s_cmp_eq_u32 %[cnt], 0")
    st.emit(f"s_cbranch_scc1 {L + 2}f")
    st.emit(f"{L + 1}:")
    body(st, 0, True, t(1), role)
    body(st, 1, True, t(2), role)
    assert set(st.pending) <= set(entry)                 # the loop head assumed at least these reads in flight: conservative
    st.emit("s_sub_u32 %[cnt], %[cnt], 1")
    st.emit("s_cmp_lg_u32 %[cnt], 0")
    st.emit(f"s_cbranch_scc1 {L + 1}b")
    st.emit(f"{L + 2}:")
    st.emit("s_cmp_eq_u32 %[rem], 0")
    st.emit(f"s_cbranch_scc1 {L + 3}f")
    body(st, 0, True, t(3), role)
    body(st, 1, False, t(4), role)
    st.emit(f"s_branch {L + 4}f")
    st.emit(f"{L + 3}:")
    st.pending = list(entry)
    body(st, 0, False, t(5), role)
    st.emit(f"{L + 4}:")


def emit():
    st = Stream()
    prologue(st)
    entry = [] if CFG["noread"] else [("K", f) for f in range(CFG["D"])]      # LDS reads in flight at every body entry
    assert st.pending == entry
    if CFG["dephase"]:
        st.emit("s_cmp_lg_u32 %[role], 0")
        st.emit("s_cbranch_scc1 50f")
    emit_loop(st, entry, "A", 10)
    if CFG["dephase"]:
        st.emit("s_branch 60f")
        st.emit("50:")
        emit_loop(st, entry, "B", 20)
        st.emit("60:")
    st.emit("s_waitcnt vmcnt(0)")
    st.emit("s_waitcnt lgkmcnt(0)")
    st.emit("s_barrier")
    st.emit("s_nop 15")
    st.emit("s_nop 15")
    return st.ins


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "..", "regione_amd", "csrc", "attn_loop_asm.inc")
    # RGN_ATTN_LOOP_SM_ASM ("static max", round 3): the caller guarantees a bound on every score (rgn_attention_bounded), so the
    # softmax needs no running max: P = exp2(S * c - m0) with a CONSTANT m0 - no tile max (25 VALU instructions + a cross-half
    # exchange per tile), no defer decision, no rescale block; everything else is the same loop
    variants = [("RGN_ATTN_LOOP_ASM", dict(D=4)), ("RGN_ATTN_LOOP_SM_ASM", dict(D=4, staticmax=True))]
    if os.environ.get("ATTN_GEN_EXPERIMENTS"):       # measurement builds only: prefetch distance 6 / 7, and two
        # timing-only ablations that compute WRONG results (no softmax VALU / no LDS fragment reads)
        variants += [("RGN_ATTN_LOOP_ASM_V1", dict(D=6)), ("RGN_ATTN_LOOP_ASM_V2", dict(D=7)),
                     ("RGN_ATTN_LOOP_ASM_V3", dict(D=4, novalu=True)), ("RGN_ATTN_LOOP_ASM_V4", dict(D=4, noread=True))]
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_attn_loop.py - do not edit.  Hand-scheduled KV loop of attention_asm_kernel.\n")
        for name, kw in variants:
            CFG.update(dict(D=4, novalu=False, noread=False, prio=False, staticmax=False, pkfma=os.environ.get("ATTN_PKFMA", "0") == "1",
                            pkadd=os.environ.get("ATTN_PKADD", "0") == "1", dot2sum=os.environ.get("ATTN_DOT2SUM", "0") == "1",
                            dephase=os.environ.get("ATTN_DEPHASE", "0") == "1",
                            dmaspread=os.environ.get("ATTN_DMA_SPREAD", "1") == "1"))
            CFG.update(kw)
            lines = emit()
            n_mfma = sum(1 for l in lines if l.startswith("v_mfma"))
            f.write(f"// {name}: {len(lines)} instructions, {n_mfma} MFMAs\n")
            f.write(f"#define {name} \\\n")
            for l in lines:
                f.write(f'    "{l}\\n\\t" \\\n')
            f.write('    ""\n')
        clob = [f'"a{n}"' for n in range(96)] + [f'"v{n}"' for n in range(32, 160)] + ['"memory"', '"scc"', '"vcc"']
        f.write("#define RGN_ATTN_LOOP_CLOBBERS " + ", ".join(clob) + "\n")
    print(f"wrote {os.path.normpath(out)}")


if __name__ == "__main__":
    main()
