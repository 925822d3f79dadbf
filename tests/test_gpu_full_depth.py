"""-m gpu: parity with the oracle at the REAL DEPTH of the trunks - tools/parity_full_depth.py
with assertions.

Rounds 1-3 showed >= 40 dB on 2-4 block trunks and on single full-width blocks; the headline configuration stacks 19 + 38
(FLUX.1-Kontext, reference FluxKontext/inplace.py:507-555) or 60 (Qwen-Image-Edit) of them.  Two cuts through that:

  * full WIDTH and full depth (d = 3072, 24 x 128, 11.9 B / 20.4 B synthetic parameters) on a 16 x 16 token grid with 64 text
    rows: one FULL step with K/V store and one REGION step (K_e = 64, fp16 round trip on the rewritten rows) - velocity and the
    last layer's K / V^T slabs >= 40 dB against the oracle's torch-CPU bf16 run, untouched cache rows bit-identical;
  * full depth at d = 512 through all 28 steps of RegionEHelper against oracle.denoise: plan and edited ids exact, final latents
    >= 40 dB for every family (trunk statistics calibrated to a checkpoint's regime, see the test's docstring).

Tolerance: 40 dB = BASELINE.json north_star "PSNR >= 40 dB vs reference latents".
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# 35-80 s of CPU oracle each (the host's load decides).  The same function at the HEADLINE shape (64 x 64 grid, T = 512, K_e = 1024) is
# test_headline_shape_vs_committed_oracle_fixture below (the oracle side, ~5 min of CPU, is a generated fixture).
@pytest.mark.gpu
@pytest.mark.parametrize("family", ["flux", "qwen"])
def test_full_depth_full_width_full_store_then_region_step_vs_oracle(family):
    import parity_full_depth as P
    r = P.full_width(family, truth=False)
    assert r["blocks"] == (57 if family == "flux" else 60) and r["d"] == 3072
    assert len(r["rows"]) == 6
    for row in r["rows"]:
        assert row["psnr_hip_vs_oracle_db"] >= 40.0 and row["rel_hip_vs_oracle"] < 5e-2, row
    assert r["untouched_rows_bit_identical"]


@pytest.mark.parametrize("family", [pytest.param("flux", marks=pytest.mark.gpu), pytest.param("qwen", marks=pytest.mark.gpu),
                                    pytest.param("step1x_v1p2", marks=pytest.mark.gpu)])
def test_full_depth_trunk_28_steps_vs_oracle_denoise(family):
    """All three families: the north star's 40 dB, hard, on a trunk with checkpoint-like statistics (round 6: AdaLN gates / scales of std 0.1,
    RMSNorm weights 1.5 +- 0.3 - tools/parity_full_depth.py CALIBRATED).  Rounds 4-5 ran gate-1 N(0, 1/d) trunks on which Qwen's 60 blocks
    + norm-preserving CFG combine `neg + 4 (pos - neg)` put even the oracle's OWN re-ordered run 39.8 dB from itself and carried a
    per-family exception; on the calibrated trunk that spread is 44 dB (profiles/r06_parity_full_depth.json, measured in the same round)
    and the exception is gone."""
    import parity_full_depth as P
    r = P.narrow_loop(family, alt=False)
    assert r["blocks"] == (60 if family == "qwen" else 57) and r["calibrated_statistics"] == P.CALIBRATED
    assert r["hip_plan"] == r["oracle_plan"] and "R" in r["hip_plan"] and "C" in r["hip_plan"]
    assert r["ids_bit_exact"] and 0 < r["hip_K_e"] < 256
    assert r["psnr_final_db"] >= 40.0, r["psnr_final_db"]
    assert torch.isfinite(torch.tensor(r["rel_final"]))


HEADLINE_FIXTURE = os.path.join(ROOT, "tests", "golden", "headline_flux.npz")


@pytest.mark.gpu
def test_headline_shape_vs_committed_oracle_fixture():
    """The shape bench.py runs - FLUX.1-Kontext 19 + 38 blocks, d = 3072, 64 x 64 grid (L = L_c = 4096), T = 512, K_e = 1024
    (reference FluxKontext/inplace.py:507-555 trunk, :694-824 processor): one FULL step with K/V store, then one REGION step with the
    partial K/V update and the fp16 round trip on the rewritten rows.  The oracle side (~5 min of CPU on 128 cores) is a committed,
    generated fixture (`python tools/parity_full_depth.py --cases flux_headline --save-fixture tests/golden/headline_flux.npz` on an
    MI355X box: both velocities in full, 64 / 32 seeded rows of the last layer's K / V^T slabs, a fingerprint of the device-drawn
    weights); the HIP side runs here.  A box whose device RNG stream gives other weights than the fixture's falls back to running
    the oracle in this process (slow, same assertions).  Tolerance: 40 dB (north_star), untouched cache rows bit-identical."""
    import parity_full_depth as P
    try:
        r = P.full_width("flux", grid=64, T=512, truth=False, alt=False, load_fixture=HEADLINE_FIXTURE)
    except P.FixtureMismatch as e:
        print(f"[headline parity] {e}: running the oracle live", flush=True)
        r = P.full_width("flux", grid=64, T=512, truth=False, alt=False)
    assert r["blocks"] == 57 and r["d"] == 3072 and r["grid"] == [64, 64] and r["T"] == 512 and r["K_e"] == 1024
    assert len(r["rows"]) == 6
    for row in r["rows"]:
        assert row["psnr_hip_vs_oracle_db"] >= 40.0 and row["rel_hip_vs_oracle"] < 5e-2, row
    assert r["untouched_rows_bit_identical"]
    print("[headline parity]", r["oracle_side"], [(x["name"], x["psnr_hip_vs_oracle_db"]) for x in r["rows"]])
