#!/usr/bin/env python3
"""Generate the hand-scheduled K loop of the 256x256x64 bf16 GEMM (regione_amd/csrc/gemm_loop_asm.inc).

    python tools/gen_gemm_loop.py            # rewrites the .inc (committed; the build does not run this script)

Why a generator: the loop is ONE inline-asm statement per variant - hipcc schedules `ds_read`, LDS-DMA and MFMA
its own way (every 8 MFMAs it waits lgkmcnt(0) in front of two fresh fragment reads, drains vmcnt(0) in front of the
tile barrier) and cannot be talked out of it.  Here every instruction slot is placed by hand:

  * 4 waves (2 x 2), wave tile 128 x 128 = 8 x 8 `v_mfma_f32_16x16x32_bf16`, 256 accumulator registers in AGPRs
    a[0:255]; fragment registers v[128:255] = two sets (one per 32-deep k-half), so the fragments of the NEXT k-half are
    always in flight behind the MFMAs of the current one (no exposed LDS latency);
  * two 64 KiB LDS stages ([A 256 rows][B 256 rows] x 128-byte rows, 16-byte slot ^= row & 7 - the same image
    gemm_bf16_kernel builds); ONE barrier per K tile, placed in the MIDDLE of the tile: when a wave arrives its second
    fragment set is already loaded, so it leaves the barrier with 64 MFMAs of work in hand; behind the barrier the
    stage just read is free and the wave's 16 LDS-DMA pieces of tile t+2 are issued one per 4 MFMAs;
  * counted waits only: lgkmcnt(0) falls 32+ MFMAs after the last read was issued, vmcnt(0) a full tile after the
    last piece was issued.

Register plan (per wave):  a[(i*TN + j)*4 + r]  accumulator (i = A fragment, j = W fragment, r = column in the lane's 4)
                           v[128 + 64*set + 4*f]  fragment f of set (f < 8: A rows, f >= 8: W rows)
Named operands (bound in gemm.hip): oa0..oa7 / ob0..ob7 per-lane source byte offsets of the A / W pieces, la0 / la1 LDS read
address of the A fragments (k-half 0 / 1), lb0 / lb1 of the W fragments, pa / pw buffer resources of A / W (4 SGPRs each),
cnt loop count (nk - 2), stg LDS base of this wave's DMA slots in the stage refilled next, koff byte offset of the current
K tile (the buffer instruction's scalar offset: ONE s_add per tile advances both operands).
"""
import os

TM = TN = 8
FRAG0 = 128                     # first fragment VGPR
A_BYTES = 256 * 128             # A region of a stage
STAGE = 2 * A_BYTES
# named asm operands (gemm.hip binds them)
PA, PW, CNT, STG, KOFF = "[pa]", "[pw]", "[cnt]", "[stg]", "[koff]"


def OA(q):
    return f"[oa{q}]"


def OB(q):
    return f"[ob{q}]"


def LA(kh):
    return f"[la{kh}]"


def LB(kh):
    return f"[lb{kh}]"


def frag(set_, f):
    b = FRAG0 + 64 * set_ + 4 * f
    return f"v[{b}:{b + 3}]"


def acc(i, j):
    b = (i * TN + j) * 4
    return f"a[{b}:{b + 3}]"


def mfma(set_, k):
    if os.environ.get("GEMM_LOOP_MFMA32") == "1":
        # TIMING-ONLY experiment (wrong arithmetic): the same pipe time as half as many 32x32x16 instructions - does the loop gain
        # from the freed issue slots?  slot k even: one 32x32x16 MFMA on accumulator block k/4 (two per block per k-half); k odd: none
        if k % 2:
            return "s_nop 0"
        blk, kk = divmod(k // 2, 2)
        b = blk * 16
        return f"v_mfma_f32_32x32x16_bf16 a[{b}:{b + 15}], {frag(set_, 8 + (blk % 4) * 2 + kk)}, {frag(set_, (blk // 4) * 2 + kk)}, a[{b}:{b + 15}]"
    i, j = divmod(k, TN)
    return f"v_mfma_f32_16x16x32_bf16 {acc(i, j)}, {frag(set_, 8 + j)}, {frag(set_, i)}, {acc(i, j)}"


def zero_acc():
    return [f"v_accvgpr_write_b32 a{n}, 0" for n in range(TM * TN * 4)]


# ------------------------------------------------------------------------------------------------------------------
# variant 0: two 64 KiB stages [A | W]; every operand has ONE tile of lead
# ------------------------------------------------------------------------------------------------------------------
def reads(set_, khalf):
    """16 fragment reads of one k-half into fragment set `set_` (A and W alternating, in the order the MFMAs need them)."""
    out = []
    for f in range(8):
        out.append(f"ds_read_b128 {frag(set_, f)}, %{LA(khalf)} offset:{f * 2048}")
        out.append(f"ds_read_b128 {frag(set_, 8 + f)}, %{LB(khalf)} offset:{A_BYTES + f * 2048}")
    return out


def dma_pieces():
    """(m0 setup, DMA) pairs of the 16 pieces one wave stages per tile."""
    out = []
    for q in range(8):
        out.append((f"s_add_u32 m0, %{STG}, {q * 1024}", f"buffer_load_dwordx4 %{OA(q)}, %{PA}, %{KOFF} offen lds"))
    for q in range(8):
        out.append((f"s_add_u32 m0, %{STG}, {A_BYTES + q * 1024}", f"buffer_load_dwordx4 %{OB(q)}, %{PW}, %{KOFF} offen lds"))
    return out


def advance():
    return [f"s_add_u32 %{KOFF}, %{KOFF}, 128", f"s_xor_b32 %{STG}, %{STG}, 0x10000"]


def prologue():
    ins = []
    for _tile in range(2):
        for m0set, ld in dma_pieces():
            ins += [m0set, "s_nop 0", ld]
        ins += advance()
    ins += zero_acc()                  # accumulators cleared while the first tiles are in flight
    ins += ["s_waitcnt vmcnt(16)", "s_barrier"]
    ins += reads(0, 0)
    ins += ["s_waitcnt lgkmcnt(0)"]
    return ins


def body(dma, nxt):
    """One K tile: phase 1 = k-half 0 (fragment set 0), phase 2 = k-half 1 (set 1)."""
    ins = []
    r1 = reads(1, 1)
    for k in range(64):
        ins.append(mfma(0, k))
        if k % 2 == 1 and r1:
            ins.append(r1.pop(0))
    assert not r1
    ins.append("s_waitcnt lgkmcnt(0)")
    if nxt:
        ins += ["s_waitcnt vmcnt(0)", "s_barrier"]
        for o in (LA(0), LA(1), LB(0), LB(1)):
            ins.append(f"v_xor_b32 %{o}, 0x10000, %{o}")
    else:
        ins.append("s_barrier")                    # last tile: every wave is done with the stages -> the epilogue may reuse LDS
    r0 = reads(0, 0) if nxt else []
    pcs = dma_pieces() if dma else []
    adv = advance() if dma else []
    for k in range(64):
        ins.append(mfma(1, k))
        g = k % 4
        if g == 0 and pcs:
            ins.append(pcs[0][0])                 # m0 of the piece issued behind the next MFMA
        elif g == 1 and pcs:
            ins.append(pcs.pop(0)[1])
        elif g == 2 and r0:
            ins.append(r0.pop(0))
        elif g == 3 and r0:
            ins.append(r0.pop(0))
        if not pcs and adv and g == 1:
            ins.append(adv.pop(0))
    ins += adv
    assert not r0 and not pcs
    if nxt:
        ins.append("s_waitcnt lgkmcnt(0)")
    return ins


def emit_v0():
    lines = []
    lines += prologue()
    lines += [f"s_cmp_eq_u32 %{CNT}, 0", "s_cbranch_scc1 2f", "1:"]
    lines += body(True, True)
    lines += [f"s_sub_u32 %{CNT}, %{CNT}, 1", f"s_cmp_lg_u32 %{CNT}, 0", "s_cbranch_scc1 1b", "2:"]
    lines += body(False, True)
    lines += body(False, False)
    lines += ["s_nop 15", "s_nop 15"]
    return lines


# ------------------------------------------------------------------------------------------------------------------
# variant 1: all 160 KiB of LDS - A ring of TWO 32 KiB slots (0, 32 K), W ring of THREE (64 K, 96 K, 128 K).  The weight
# operand is the HBM-cold one (every W panel is read by one or two XCDs, straight from HBM; the activations were just
# written by the previous kernel and sit in L2 / Infinity Cache), so W gets TWO tiles of lead, A one.  In the middle of
# tile t (behind the barrier) the wave issues A(t+2) into the A slot just freed and W(t+3) into the W slot just freed; the
# wait in front of the barrier is `vmcnt(8)`: everything but the 8 pieces of W(t+2) must have landed.
# extra operands: was (sgpr) LDS base of the W slot refilled next (this wave's pieces), wrd (sgpr) LDS base of the W slot
# read next, lbo0 / lbo1 (vgpr) offsets of the W fragments inside a slot; stg = the wave's pieces in the A slot refilled next
# ------------------------------------------------------------------------------------------------------------------
WAS, WRD, KOFW = "[was]", "[wrd]", "[kofw]"
W_SLOT0, W_END, SLOT = 65536, 163840, 32768


def LBO(kh):
    return f"[lbo{kh}]"


ABL = os.environ.get("GEMM_LOOP_ABL", "")       # TIMING-ONLY ablations (wrong results): "noA" = no A pieces, no A fragment reads;
#                                                   "noA+ld" = the same plus 16 direct global loads of A fragments per K tile


def reads1(set_, khalf):
    out = []
    for f in range(8):
        if ABL == "noA+ld":
            out.append(f"s_sub_u32 m0, %{KOFF}, 256\\n\\ts_nop 0\\n\\tbuffer_load_dwordx4 {frag(set_, f)}, %{OA(f)}, %{PA}, m0 offen offset:{khalf * 64}")
        elif ABL == "noA":
            out.append("s_nop 0")
        else:
            out.append(f"ds_read_b128 {frag(set_, f)}, %{LA(khalf)} offset:{f * 2048}")
        out.append(f"ds_read_b128 {frag(set_, 8 + f)}, %{LB(khalf)} offset:{f * 2048}")
    return out


def dma_a():
    if ABL:
        return [("s_nop 0", "s_nop 0") for q in range(8)]
    return [(f"s_add_u32 m0, %{STG}, {q * 1024}", f"buffer_load_dwordx4 %{OA(q)}, %{PA}, %{KOFF} offen lds") for q in range(8)]


def dma_w():
    return [(f"s_add_u32 m0, %{WAS}, {q * 1024}", f"buffer_load_dwordx4 %{OB(q)}, %{PW}, %{KOFW} offen lds") for q in range(8)]


def rot(reg):
    """advance an LDS W-slot base by one slot, wrapping after the third"""
    return [f"s_add_u32 %{reg}, %{reg}, {SLOT}", f"s_cmp_lt_u32 %{reg}, {W_END}", f"s_cselect_b32 %{reg}, %{reg}, %[wwrap]"]


def rot_was():
    # the wrap target of `was` is slot 0 + this wave's piece offset = %[wwrap]; `wrd` wraps to W_SLOT0 (no wave offset)
    return [f"s_add_u32 %{WAS}, %{WAS}, {SLOT}", f"s_cmp_lt_u32 %{WAS}, {W_END}", f"s_cselect_b32 %{WAS}, %{WAS}, %[wwrap]"]


def rot_wrd():
    return [f"s_add_u32 %{WRD}, %{WRD}, {SLOT}", f"s_cmp_lt_u32 %{WRD}, {W_END}", f"s_cselect_b32 %{WRD}, %{WRD}, {W_SLOT0}"]


CONV = False      # emit_v3: the A operand walks a 3 x 3 convolution window (see emit_v3)


def adv_a():
    if CONV:
        # K runs over (ky, kx, c): inside one kernel row the three taps' channels are CONTIGUOUS in a pixel-major [rows, C] image (pixel
        # x+1 follows pixel x), so the A offset advances by 128 bytes per K tile as in a plain GEMM; after the 3C/64 tiles of a kernel
        # row it jumps to the next image row: + cjump bytes.  ctap = K tiles left in the current kernel row, crow = 3C/64.
        return [f"s_add_u32 %{KOFF}, %{KOFF}, 128", "s_sub_u32 %[ctap], %[ctap], 1", "s_cmp_eq_u32 %[ctap], 0",
                "s_cselect_b32 %[tj], %[cjump], 0", "s_cselect_b32 %[ctap], %[crow], %[ctap]", f"s_add_u32 %{KOFF}, %{KOFF}, %[tj]",
                f"s_xor_b32 %{STG}, %{STG}, 0x8000"]
    return [f"s_add_u32 %{KOFF}, %{KOFF}, 128", f"s_xor_b32 %{STG}, %{STG}, 0x8000"]


def adv_w():
    return [f"s_add_u32 %{KOFW}, %{KOFW}, 128"] + rot_was()


def prologue1():
    ins = []

    def issue(pcs):
        for m0set, ld in pcs:
            ins.extend([m0set, "s_nop 0", ld])
    issue(dma_a()); ins.extend(adv_a())          # A(0)
    issue(dma_w()); ins.extend(adv_w())          # W(0)
    issue(dma_a()); ins.extend(adv_a())          # A(1)
    issue(dma_w()); ins.extend(adv_w())          # W(1)
    issue(dma_w()); ins.extend(adv_w())          # W(2)
    ins += zero_acc()
    ins += ["s_waitcnt vmcnt(24)", "s_barrier"]
    ins += [f"v_add_u32 %{LB(0)}, %{WRD}, %{LBO(0)}", f"v_add_u32 %{LB(1)}, %{WRD}, %{LBO(1)}"]
    ins += reads1(0, 0)
    ins += ["s_waitcnt lgkmcnt(0)"]
    return ins


def body1(dmaa, dmaw, nxt, wait):
    ins = []
    r1 = reads1(1, 1)
    for k in range(64):
        ins.append(mfma(0, k))
        if k % 2 == 1 and r1:
            ins.append(r1.pop(0))
    ins.append("s_waitcnt lgkmcnt(0)")
    if nxt:
        ins += [f"s_waitcnt vmcnt({wait})", "s_barrier"]
        ins += rot_wrd()
        ins += [f"v_xor_b32 %{LA(0)}, 0x8000, %{LA(0)}", f"v_xor_b32 %{LA(1)}, 0x8000, %{LA(1)}",
                f"v_add_u32 %{LB(0)}, %{WRD}, %{LBO(0)}", f"v_add_u32 %{LB(1)}, %{WRD}, %{LBO(1)}"]
    else:
        ins.append("s_barrier")
    r0 = reads1(0, 0) if nxt else []
    pcs = (dma_a() if dmaa else []) + (dma_w() if dmaw else [])
    adv = (adv_a() if dmaa else []) + (adv_w() if dmaw else [])
    for k in range(64):
        ins.append(mfma(1, k))
        g = k % 4
        if g == 0 and pcs:
            ins.append(pcs[0][0])
        elif g == 1 and pcs:
            ins.append(pcs.pop(0)[1])
        elif g == 2 and r0:
            ins.append(r0.pop(0))
        elif g == 3 and r0:
            ins.append(r0.pop(0))
    ins += adv
    assert not r0 and not pcs
    if nxt:
        ins.append("s_waitcnt lgkmcnt(0)")
    return ins


def body1e(dmaa, dmaw, nxt, wait):
    """Round 3 (the shipped ring variant; GEMM_LOOP_EARLY_BARRIER=0 regenerates the round-2 schedule `body1`): the tile barrier in
    the MIDDLE of phase 1 instead of at the phase boundary.  The 16 set-1 reads go out in the first 16 MFMA slots; at slot BAR
    every wave has its tile-t fragments in registers, so the stage is free there already: the 8 A pieces of tile t+2 follow in the
    rest of phase 1 (one instruction per two MFMAs), the 8 W pieces of tile t+3 are spread over the whole of phase 2 (one per eight
    MFMAs) next to the 16 reads of the next tile (one per two MFMAs, first half) - half the LDS-DMA pieces per phase, none next to
    the set-1 reads, never two DMA-related instructions in adjacent slots.  An LDS-DMA piece costs the issuing wave ~60 cycles among
    bare MFMAs but 100-185 in a phase that already carries 8 pieces + 16 ds_read_b128 (MI355X_MICROARCH.md), and a wave of this
    kernel is the only one on its SIMD: every cycle it spends issuing is a cycle the MFMA pipe may run dry.  Same MFMAs, same order:
    bit-identical accumulators.  Measured, same box, cold weights (2 boxes): proj_out (K = 15360) 1268 -> 1357 TFLOP/s, ff2
    (K = 12288) 1340 -> 1430, K = 3072 shapes +-1 %; in the pipeline every shape gains (kvq+mlp 1240 -> 1262, proj_out 1304 -> 1377,
    ff2 1201 -> 1311): 20.14 -> 20.52 steps/s (+1.9 %).  Variants measured next to it: all 16 pieces packed into phase 1 behind
    the barrier (one per MFMA) -6 %; barrier at slot 16 / 24 / 28, reads-first phase 2: within +-0.3 % of this one."""
    BAR = int(os.environ.get("GEMM_LOOP_BAR", "20"))
    WPH1 = os.environ.get("GEMM_LOOP_WPH1") == "1"          # the W pieces in phase 1 too (phase 2 then carries reads only)
    STEP = int(os.environ.get("GEMM_LOOP_STEP", "2"))       # MFMA slots per DMA-related instruction in phase 1
    ins = []
    r1 = reads1(1, 1)
    pa = (dma_a() if dmaa else []) + ((dma_w() if dmaw else []) if WPH1 else [])
    flat = [x for pr in pa for x in pr]
    for k in range(64):
        ins.append(mfma(0, k))
        if k < 16:
            ins.append(r1.pop(0))
        if k == BAR:
            ins.append("s_waitcnt lgkmcnt(0)")
            if nxt:
                ins += [f"s_waitcnt vmcnt({wait + 24 if ABL == 'noA+ld' and wait else wait})", "s_barrier"]
                ins += rot_wrd()
                ins += [f"v_xor_b32 %{LA(0)}, 0x8000, %{LA(0)}", f"v_xor_b32 %{LA(1)}, 0x8000, %{LA(1)}",
                        f"v_add_u32 %{LB(0)}, %{WRD}, %{LBO(0)}", f"v_add_u32 %{LB(1)}, %{WRD}, %{LBO(1)}"]
            else:
                ins.append("s_barrier")
        if k > BAR and flat and (k - BAR - 1) % STEP == 0:
            ins.append(flat.pop(0))
    assert not flat, len(flat)
    if dmaa:
        ins += adv_a()
    if dmaw and WPH1:
        ins += adv_w()
    r0 = reads1(0, 0) if nxt else []
    pw = (dma_w() if dmaw else []) if not WPH1 else []
    flatw = [x for pr in pw for x in pr]
    P2 = os.environ.get("GEMM_LOOP_P2", "spread")
    for k in range(64):
        ins.append(mfma(1, k))
        g = k % 4
        if P2 == "a":                               # reads and W pieces interleaved in the first half
            if g in (2, 3) and r0:
                ins.append(r0.pop(0))
            elif g in (0, 1) and flatw and k >= 2:
                ins.append(flatw.pop(0))
        elif P2 == "spread":                        # reads one per two slots in the first half, W pieces one per eight slots over the phase
            if k % 2 == 0 and r0:
                ins.append(r0.pop(0))
            if k % 8 in (5, 7) and flatw:
                ins.append(flatw.pop(0))
        elif P2 == "spread2":                       # 16 reads in the first 16 slots, W pieces one per eight slots over the whole phase
            if k < 16 and r0:
                ins.append(r0.pop(0))
            if k % 8 in (5, 7) and flatw:
                ins.append(flatw.pop(0))
        elif P2 == "readsfirst":                    # 16 reads in the first 16 slots, then the W pieces (one instruction per 3 slots)
            if k < 16 and r0:
                ins.append(r0.pop(0))
            if k >= 16 and (k - 16) % 3 == 0 and flatw:
                ins.append(flatw.pop(0))
    assert not r0 and not flatw, (len(r0), len(flatw))
    if dmaw and not WPH1:
        ins += adv_w()
    if nxt:
        ins.append("s_waitcnt lgkmcnt(0)")
    return ins


def emit_v1():
    """cnt = nk - 3 full bodies (t = 0 .. nk-4), then t = nk-3 (A only), nk-2 (no DMA), nk-1 (last).  Needs nk >= 4."""
    if os.environ.get("GEMM_LOOP_EARLY_BARRIER", "1") == "1":
        lines = prologue1()
        lines += ["1:"]
        lines += body1e(True, True, True, 8)
        lines += [f"s_sub_u32 %{CNT}, %{CNT}, 1", f"s_cmp_lg_u32 %{CNT}, 0", "s_cbranch_scc1 1b"]
        lines += body1e(True, False, True, 8)
        lines += body1e(False, False, True, 0)
        lines += body1e(False, False, False, 0)
        lines += ["s_nop 15", "s_nop 15"]
        return lines
    lines = prologue1()
    lines += ["1:"]
    lines += body1(True, True, True, 8)
    lines += [f"s_sub_u32 %{CNT}, %{CNT}, 1", f"s_cmp_lg_u32 %{CNT}, 0", "s_cbranch_scc1 1b"]
    lines += body1(True, False, True, 8)
    lines += body1(False, False, True, 0)
    lines += body1(False, False, False, 0)
    lines += ["s_nop 15", "s_nop 15"]
    return lines


# ------------------------------------------------------------------------------------------------------------------
# variant 2: fp8 (OCP e4m3fn) WEIGHTS in the ring variant's loop.  The W tile travels and sits in LDS as BYTES: 256 rows x
# 64 B = 16 KiB per K tile (half of the bf16 image), 4 DMA pieces per wave instead of 8, a W ring of three 16 KiB slots at
# 64 K / 80 K / 96 K (two tiles of lead, like variant 1).  A fragment is one `ds_read_b64` (8 fp8 values of one row) into
# the UPPER two registers of the fragment's four, widened in place by four `v_cvt_scalef32_pk_bf16_fp8` (scale 1.0: exact,
# every e4m3 value is a bf16 value): f0 <- lo(r0), f1 <- hi(r0), f2 <- lo(r1), f3 <- hi(r1) with (r0, r1) = (f2, f3) - no
# extra registers.  The MFMAs, their order and the A side are variant 1's, so the accumulators are bit-identical to the
# compiler-scheduled fp8-tile kernel (gemm_bf16_kernel<.., AV = 8>) and to the widen-once path.
# Per phase the W raw reads go FIRST (LDS returns in order: `lgkmcnt(8)` = the eight raw reads have landed while the eight A
# reads may still fly), then one convert per MFMA slot.
# LDS image of a W slot: piece p (1 KiB) = rows 16p .. 16p+15, lane l -> row l >> 2, 16-byte chunk l & 3 (the source chunk is
# swizzled ^ ((row >> 2) & 3), the 8-byte fragment slot ^ (((row >> 2) & 3) << 1) - the AV = 8 kernel's image).
# operands: ob0..ob3 (4 W pieces per wave), kofw advances by 64 bytes per K tile; everything else as variant 1.
# ------------------------------------------------------------------------------------------------------------------
W8_SLOT, W8_END = 16384, 65536 + 3 * 16384


def raw(set_, f):
    b = FRAG0 + 64 * set_ + 4 * (8 + f)
    return f"v[{b + 2}:{b + 3}]"


def reads8_w(set_, khalf):
    return [f"ds_read_b64 {raw(set_, f)}, %{LB(khalf)} offset:{f * 1024}" for f in range(8)]


def reads8_a(set_, khalf):
    return [f"ds_read_b128 {frag(set_, f)}, %{LA(khalf)} offset:{f * 2048}" for f in range(8)]


def cvts(set_):
    out = []
    for f in range(8):
        b = FRAG0 + 64 * set_ + 4 * (8 + f)
        out += [f"v_cvt_scalef32_pk_bf16_fp8 v{b}, v{b + 2}, 1.0",
                f"v_cvt_scalef32_pk_bf16_fp8 v{b + 1}, v{b + 2}, 1.0 op_sel:[1,0,0]",
                f"v_cvt_scalef32_pk_bf16_fp8 v{b + 2}, v{b + 3}, 1.0",
                f"v_cvt_scalef32_pk_bf16_fp8 v{b + 3}, v{b + 3}, 1.0 op_sel:[1,0,0]"]
    return out


def dma_w8():
    return [(f"s_add_u32 m0, %{WAS}, {q * 1024}", f"buffer_load_dwordx4 %{OB(q)}, %{PW}, %{KOFW} offen lds") for q in range(4)]


def adv_w8():
    return [f"s_add_u32 %{KOFW}, %{KOFW}, 64", f"s_add_u32 %{WAS}, %{WAS}, {W8_SLOT}", f"s_cmp_lt_u32 %{WAS}, {W8_END}",
            f"s_cselect_b32 %{WAS}, %{WAS}, %[wwrap]"]


def rot_wrd8():
    return [f"s_add_u32 %{WRD}, %{WRD}, {W8_SLOT}", f"s_cmp_lt_u32 %{WRD}, {W8_END}", f"s_cselect_b32 %{WRD}, %{WRD}, {W_SLOT0}"]


def prologue2():
    ins = []

    def issue(pcs):
        for m0set, ld in pcs:
            ins.extend([m0set, "s_nop 0", ld])
    issue(dma_a()); ins.extend(adv_a())          # A(0)
    issue(dma_w8()); ins.extend(adv_w8())        # W(0)
    issue(dma_a()); ins.extend(adv_a())          # A(1)
    issue(dma_w8()); ins.extend(adv_w8())        # W(1)
    issue(dma_w8()); ins.extend(adv_w8())        # W(2)
    ins += zero_acc()
    ins += ["s_waitcnt vmcnt(16)", "s_barrier"]  # all but A(1) 8 + W(1) 4 + W(2) 4
    ins += [f"v_add_u32 %{LB(0)}, %{WRD}, %{LBO(0)}", f"v_add_u32 %{LB(1)}, %{WRD}, %{LBO(1)}"]
    ins += reads8_w(0, 0) + reads8_a(0, 0)
    ins += ["s_waitcnt lgkmcnt(8)"] + cvts(0) + ["s_waitcnt lgkmcnt(0)"]
    return ins


def phase(set_mfma, set_next, khalf_next, pcs, adv, load_next):
    """64 MFMAs on fragment set `set_mfma`; behind them: the raw W / A reads of (set_next, khalf_next), their converts, and the
    DMA pieces `pcs` (m0 set in slot k, load in slot k + 1)."""
    ins = []
    rw = reads8_w(set_next, khalf_next) if load_next else []
    ra = reads8_a(set_next, khalf_next) if load_next else []
    cv = cvts(set_next) if load_next else []
    pcs, adv = list(pcs), list(adv)
    flat = [x for pr in pcs for x in pr]                 # m0 set, load, m0 set, load ...
    for k in range(64):
        ins.append(mfma(set_mfma, k))
        if k < 8 and rw:
            ins.append(rw.pop(0))
        elif 8 <= k < 16 and ra:
            ins.append(ra.pop(0))
        if flat and k < 48:
            ins.append(flat.pop(0))
        if k == 23 and load_next:
            ins.append("s_waitcnt lgkmcnt(8)")           # the eight raw W reads (issued first) have landed
        if 24 <= k < 56 and cv:
            ins.append(cv.pop(0))
    assert not rw and not ra and not cv and not flat
    ins += adv
    if load_next:
        ins.append("s_waitcnt lgkmcnt(0)")
    return ins


def body2(dmaa, dmaw, nxt, wait):
    ins = phase(0, 1, 1, [], [], True)
    if nxt:
        ins += [f"s_waitcnt vmcnt({wait})", "s_barrier"]
        ins += rot_wrd8()
        ins += [f"v_xor_b32 %{LA(0)}, 0x8000, %{LA(0)}", f"v_xor_b32 %{LA(1)}, 0x8000, %{LA(1)}",
                f"v_add_u32 %{LB(0)}, %{WRD}, %{LBO(0)}", f"v_add_u32 %{LB(1)}, %{WRD}, %{LBO(1)}"]
    else:
        ins.append("s_barrier")
    pcs = (dma_a() if dmaa else []) + (dma_w8() if dmaw else [])
    adv = (adv_a() if dmaa else []) + (adv_w8() if dmaw else [])
    ins += phase(1, 0, 0, pcs, adv, nxt)
    return ins


def body2e(dmaa, dmaw, nxt, wait):
    """The fp8 loop on the early-barrier schedule of body1e: barrier at slot BAR8 of phase 1 (every read of tile t - raw W and A -
    has landed), the converts of set 1 and the 8 A pieces behind it, the 4 W pieces spread over phase 2."""
    BAR = int(os.environ.get("GEMM_LOOP_BAR8", "24"))
    ins = []
    rw, ra, cv = reads8_w(1, 1), reads8_a(1, 1), cvts(1)
    flat = [x for pr in (dma_a() if dmaa else []) for x in pr]
    for k in range(64):
        ins.append(mfma(0, k))
        if k < 8:
            ins.append(rw.pop(0))
        elif k < 16:
            ins.append(ra.pop(0))
        if k == BAR:
            ins.append("s_waitcnt lgkmcnt(0)")
            if nxt:
                ins += [f"s_waitcnt vmcnt({wait})", "s_barrier"]
                ins += rot_wrd8()
                ins += [f"v_xor_b32 %{LA(0)}, 0x8000, %{LA(0)}", f"v_xor_b32 %{LA(1)}, 0x8000, %{LA(1)}",
                        f"v_add_u32 %{LB(0)}, %{WRD}, %{LBO(0)}", f"v_add_u32 %{LB(1)}, %{WRD}, %{LBO(1)}"]
            else:
                ins.append("s_barrier")
        if k > BAR and cv:
            ins.append(cv.pop(0))
        if k > BAR and flat and (k - BAR - 1) % 2 == 0:
            ins.append(flat.pop(0))
    assert not cv and not flat and not rw and not ra
    if dmaa:
        ins += adv_a()
    rw, ra, cv = (reads8_w(0, 0), reads8_a(0, 0), cvts(0)) if nxt else ([], [], [])
    flatw = [x for pr in (dma_w8() if dmaw else []) for x in pr]
    for k in range(64):
        ins.append(mfma(1, k))
        if k < 8 and rw:
            ins.append(rw.pop(0))
        elif 8 <= k < 16 and ra:
            ins.append(ra.pop(0))
        if k % 8 in (5, 7) and flatw:
            ins.append(flatw.pop(0))
        if k == 23 and nxt:
            ins.append("s_waitcnt lgkmcnt(8)")           # the eight raw W reads (issued first) have landed
        if 24 <= k < 56 and cv:
            ins.append(cv.pop(0))
    assert not rw and not ra and not cv and not flatw
    if dmaw:
        ins += adv_w8()
    if nxt:
        ins.append("s_waitcnt lgkmcnt(0)")
    return ins


def emit_v2():
    """cnt = nk - 3 full bodies, then t = nk-3 (A only), nk-2 (no DMA), nk-1 (last).  Needs nk >= 4."""
    b2 = body2e if os.environ.get("GEMM_LOOP_EARLY_BARRIER", "1") == "1" else body2
    lines = prologue2()
    lines += ["1:"]
    lines += b2(True, True, True, 4)
    lines += [f"s_sub_u32 %{CNT}, %{CNT}, 1", f"s_cmp_lg_u32 %{CNT}, 0", "s_cbranch_scc1 1b"]
    lines += b2(True, False, True, 4)
    lines += b2(False, False, True, 0)
    lines += b2(False, False, False, 0)
    lines += ["s_nop 15", "s_nop 15"]
    return lines


def emit_v3():
    """The ring variant (emit_v1) as an IMPLICIT-GEMM 3 x 3 convolution over a zero-bordered, pixel-major activation image [rows, C]
    (row = y * Wp + x of the PADDED image): output row m, tap (ky, kx) reads image row m + (ky - 1) * Wp + (kx - 1) - a constant row
    shift per tap, so the A tile of a K step is the plain GEMM's A tile at another byte offset.  The kernel passes A - (Wp + 1) rows
    as the base; K = 9 C in (ky, kx, c) order; extra operands: ctap (sgpr, in/out: K tiles left in the current kernel row), crow
    (sgpr: 3C/64), cjump (sgpr: (Wp - 3) * C * 2 bytes), tj (sgpr scratch).  Same MFMAs in the same order as emit_v1."""
    global CONV
    CONV = True
    try:
        return emit_v1()
    finally:
        CONV = False


def write_macro(f, name, lines):
    n_mfma = sum(1 for l in lines if l.startswith("v_mfma"))
    f.write(f"// {name}: {len(lines)} instructions, {n_mfma} MFMAs\n")
    f.write(f"#define {name} \\\n")
    for l in lines:
        f.write(f'    "{l}\\n\\t" \\\n')
    f.write('    ""\n')


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "..", "regione_amd", "csrc", "gemm_loop_asm.inc")
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_gemm_loop.py - do not edit.  Hand-scheduled K loops of gemm_bf16_kernel<.., 256, 256, 2, 2, ..>.\n")
        write_macro(f, "RGN_GEMM_LOOP4W_ASM", emit_v0())
        write_macro(f, "RGN_GEMM_LOOP4W_RING_ASM", emit_v1())
        write_macro(f, "RGN_GEMM_LOOP4W_W8_ASM", emit_v2())
        write_macro(f, "RGN_GEMM_LOOP4W_CONV_ASM", emit_v3())
        clob = [f'"a{n}"' for n in range(256)] + [f'"v{n}"' for n in range(FRAG0, 256)] + ['"memory"', '"scc"']
        f.write("#define RGN_GEMM_LOOP4W_CLOBBERS " + ", ".join(clob) + "\n")
    print(f"wrote {os.path.normpath(out)}")


if __name__ == "__main__":
    main()
