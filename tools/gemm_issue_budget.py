#!/usr/bin/env python3
"""Cycle budget of the hand-scheduled GEMM K loop and of a whole tile (where the GEMM's 0.60 MFMA-busy goes).

Parses the steady-state body of `RGN_GEMM_LOOP4W_RING_ASM` (regione_amd/csrc/gemm_loop_asm.inc: one K tile of 64 per iteration, one wave of the
4-wave workgroup = one wave per SIMD), prices it like tools/attn_issue_budget.py, and sets it against the measured figures of DESIGN 4.2 / 5:

    v_mfma_f32_16x16x32_bf16   4 passes x 4 cycles = 16 cycles of the matrix pipe (16384 FLOP at 1024 FLOP / clk / SIMD), 4 issue cycles
    ds_read_b128               4 issue cycles, 4 LDS-array cycles per wave-instruction (256 B/clk/CU)
    buffer_load ... lds        4 issue cycles (+ 60-185 cycles of address work on the issuing wave: MI355X_MICROARCH.md)

    python tools/gemm_issue_budget.py [--clock-mhz 1956] [--ktile-us 1.27] [--tile-fixed-us 6.5] > profiles/r06_gemm_issue_budget.txt
"""
import argparse
import collections
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def body(macro):
    txt = open(os.path.join(ROOT, "regione_amd", "csrc", "gemm_loop_asm.inc")).read()
    m = re.search(r"#define " + macro + r" \\\n(.*?)\n    \"\"", txt, re.S)
    lines = [l.strip()[1:].split("\\n")[0] for l in m.group(1).split("\n") if l.strip().startswith('"')]
    start = next(i for i, l in enumerate(lines) if re.match(r"^\d+:$", l))
    end = next(i for i, l in enumerate(lines) if i > start and l.startswith("s_cbranch") and l.endswith("b"))
    return lines[start + 1:end + 1]


def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("buffer_load"):
        return "vmem_dma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    return "salu"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clock-mhz", type=float, default=1956.0, help="sustained clock of the kvq+mlp GEMM (profiles/r06_clock_probe.json: 1972; in the edit 1936)")
    ap.add_argument("--ktile-us", type=float, default=1.27, help="measured time per K tile of 64 and round at full occupancy (DESIGN 4.2: 0.0194-0.0202 us per K)")
    ap.add_argument("--tile-fixed-us", type=float, default=6.5, help="measured fixed cost per tile, prologue + epilogue (DESIGN 4.2: 5-8 us)")
    ns = ap.parse_args()
    for macro, label in (("RGN_GEMM_LOOP4W_RING_ASM", "bf16 ring loop (every whole-K tile of the edit)"),
                         ("RGN_GEMM_LOOP4W_CONV_ASM", "the same loop walking a 3 x 3 convolution window (VAE)")):
        b = body(macro)
        c = collections.Counter(classify(i) for i in b)
        print(f"## {macro}: {label}")
        print(f"loop body = {len(b)} instructions per K tile of 64 and wave (wave tile 128 x 128; one wave per SIMD)")
        for k in ("mfma", "lds", "vmem_dma", "valu", "salu", "waitcnt", "barrier"):
            print(f"  {k:<10} {c.get(k, 0):5d}")
        mat = c["mfma"] * 16
        issue_vec = (c["mfma"] + c["lds"] + c["vmem_dma"] + c.get("valu", 0)) * 4
        issue_all = issue_vec + (c.get("salu", 0) + c.get("waitcnt", 0) + c.get("barrier", 0)) * 4
        lds = c["lds"] * 4 * 4 + 2 * 32768 // 128           # 4 waves' reads through the CU's array + 64 KiB of LDS-DMA writes per K tile
        cyc = ns.ktile_us * ns.clock_mhz
        print(f"per SIMD and K tile: matrix pipe {mat} cycles, vector issue {issue_vec}, all issue {issue_all}, LDS array (CU-wide) {lds}; measured {cyc:.0f} cycles "
              f"({ns.ktile_us} us at {ns.clock_mhz:.0f} MHz) -> MFMA busy INSIDE the K loop {mat / cyc:.2f}")
        print()
    # a whole launch: kvq+mlp of a FLUX single block
    M, N, K = 8704, 21504, 3072
    tiles = (M // 256) * (N // 256)
    rounds = -(-tiles // 256)
    nk = K // 64
    loop = nk * ns.ktile_us
    t_launch = rounds * (loop + ns.tile_fixed_us)
    ideal = tiles / 256.0 * nk * (128 * 16 / ns.clock_mhz)
    print(f"## a whole launch: kvq+mlp of a FLUX single block ({M} x {N} x {K}: {tiles} tiles = {tiles / 256:.2f} rounds -> {rounds})")
    print(f"  matrix-pipe time at this clock      {ideal:7.1f} us   (tiles / 256 x {nk} K tiles x 2048 cycles)")
    print(f"  x K-loop efficiency                 {tiles / 256.0 * loop:7.1f} us   ({128 * 16 / (ns.ktile_us * ns.clock_mhz):.2f}: LDS-DMA address work, barrier, waits)")
    print(f"  x round quantisation                {rounds * loop:7.1f} us   ({tiles / 256 / rounds:.2f}: the last round holds {tiles - (rounds - 1) * 256} of 256 tiles)")
    print(f"  + prologue / epilogue per tile      {t_launch:7.1f} us   (+ {ns.tile_fixed_us} us x {rounds} rounds; measured 791 us with the bias epilogue, 818-874 with the fused Q/K/V + GELU one)")
    print(f"  => MFMA busy {ideal / t_launch:.2f} of the launch (PMC: bias 0.70, fused Q/K/V 0.61); x clock {ns.clock_mhz:.0f} / 2400 MHz = {ideal / t_launch * ns.clock_mhz / 2400:.2f} of the 2.5 PFLOP/s headline peak")
    print()
    print("Reading: of the 40 % of matrix-pipe time the GEMM family does not use, ~18 points are inside the K loop (one wave per SIMD: every LDS-DMA piece,")
    print("ds_read and wait of that wave is a slot in which its SIMD's matrix pipe can run dry - the loop was re-scheduled three times, rounds 2-4), ~7 are round")
    print("quantisation (tiles of equal length on 256 CUs), ~6-10 the per-tile prologue + epilogue (VALU-bound in the fused Q/K/V form), the rest launch")
    print("remainders (split pieces 0.57, 128-geometry 0.33).  What converts busy into the headline fraction is the clock: the loop draws the board's")
    print("1.4 kW at 1.94-1.97 GHz (0.94 pJ/FLOP; hipBLASLt 0.97 at 1.93 GHz) - frac 0.51 = busy 0.60 x 1.95 / 2.4 x (launched / algorithmic FLOPs ~ 1.04).")


if __name__ == "__main__":
    main()
