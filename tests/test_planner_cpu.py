"""The GEMM launch planner's decisions for the region-step shapes, on the CPU (rgn_gemm_plan_query: pure host arithmetic on the cost
model of csrc/gemm.hip - no launch).  Pins what tools/probes/plan_sweep.py measured on MI355X (profiles/r04_plan_sweep*.txt): which
geometry / how many K pieces / quarter-tile remainder each (M, N, K) of a FLUX / Qwen / Step1X region step gets.  A cost-model edit
that flips one of these has to come with a new sweep."""
import ctypes as C

import pytest

from regione_amd import _lib

WS = 256 << 20


def plan(Ms, N, K, distinct_w=None, w8=0, ws=WS):
    arr = (C.c_int * len(Ms))(*Ms)
    p = _lib.lib().rgn_gemm_plan_query(arr, len(Ms), N, K, distinct_w or (1 if len(Ms) == 1 else 2), w8, ws)
    assert p >= 0, p
    return dict(big=bool(p & 0x400), pieces=p & 0xff, quarter=bool(p & 0x100))


@pytest.fixture(autouse=True)
def no_overrides():
    _lib.lib().rgn_plan_override(None, 0)          # every knob at -1: the cost models decide
    yield
    _lib.lib().rgn_plan_override(None, 0)


@pytest.mark.parametrize("Ms,N,K", [((1137,), 3072, 15360), ((1056,), 3072, 15360), ((1248,), 3072, 15360),
                                    ((625, 512), 3072, 12288), ((448, 512), 3072, 12288), ((736, 512), 3072, 12288)])
def test_long_k_projections_at_ke_8_to_18_percent_take_256_tiles_with_split_k(Ms, N, K):
    """48-60 tiles x K 12288 / 15360: the 128 geometry (216-240 blocks, two per CU) and the quarter-tile remainder built on it cost
    175-215 us, 256 x 256 + 3-5 K pieces 90-120 us (round-3 planner took the former)."""
    p = plan(Ms, N, K)
    assert p["big"] and not p["quarter"] and 3 <= p["pieces"] <= 5, p


@pytest.mark.parametrize("Ms,N,K,want", [
    ((1536,), 21504, 3072, dict(big=True, pieces=1, quarter=False)),          # FLUX R kvq+mlp: 504 tiles = 1.97 rounds, plain
    ((1536,), 3072, 15360, dict(big=True, pieces=3, quarter=False)),          # FLUX R proj_out: 72 tiles, 3 pieces
    ((1024, 512), 3072, 12288, dict(big=True, pieces=3, quarter=False)),      # FLUX R FF-down pair
    ((1024, 512), 9216, 3072, dict(big=True, pieces=1, quarter=False)),       # FLUX R Q/K/V pair: 216 tiles in one round
    ((1024, 1024, 512, 384), 3072, 12288, dict(big=True, pieces=1, quarter=False)),   # Qwen batched FF-down: 144 tiles, plain (213 vs 225 us)
    ((1024, 1024, 512, 384), 3072, 3072, dict(big=True, pieces=1, quarter=False)),    # Qwen batched out-projection: plain
    ((2025, 512), 3072, 12288, dict(big=True, pieces=2, quarter=False)),      # FLUX R 50 % FF-down: 120 tiles, 2 pieces
])
def test_planner_choices_for_the_headline_region_shapes(Ms, N, K, want):
    assert plan(Ms, N, K) == want


def test_full_step_projection_keeps_whole_rounds_and_splits_the_remainder():
    # proj_out of a full step: 408 tiles = one whole round + 152 -> the remainder is cut along K; kvq+mlp (2856 = 11 rounds + 40): the
    # 40-tile remainder is NOT worth two extra launches
    p = plan((8704,), 3072, 15360)
    assert p["big"] and p["pieces"] >= 2 and not p["quarter"]
    assert plan((8704,), 21504, 3072) == dict(big=True, pieces=1, quarter=False)


def test_no_workspace_means_no_split_and_overrides_are_honoured():
    assert plan((1536,), 3072, 15360, ws=0)["pieces"] <= 1          # no partials without a workspace (here: the 128 geometry instead)
    assert plan((8704,), 3072, 15360, ws=0) == dict(big=True, pieces=1, quarter=False)
    with _lib.plan_override(gemm_geometry=128):
        assert not plan((1536,), 3072, 15360)["big"]
    with _lib.plan_override(gemm_geometry=256, gemm_pieces=5):
        assert plan((1536,), 3072, 15360) == dict(big=True, pieces=5, quarter=False)
    with _lib.plan_override(gemm_pieces=1):
        assert plan((8704,), 3072, 15360) == dict(big=True, pieces=1, quarter=False)
    assert plan((1536,), 3072, 15360) == dict(big=True, pieces=3, quarter=False)          # the knobs are back at -1
    assert _lib.lib().rgn_plan_override(b"no_such_knob", 1) < 0


def test_bad_arguments_are_refused():
    arr = (C.c_int * 1)(100)
    assert _lib.lib().rgn_gemm_plan_query(arr, 1, 3072, 100, 1, 0, WS) < 0          # K not a multiple of 64
    assert _lib.lib().rgn_gemm_plan_query(arr, 5, 3072, 128, 1, 0, WS) < 0          # more than four problems
    assert _lib.lib().rgn_gemm_plan_query(None, 1, 3072, 128, 1, 0, WS) < 0


# ---------------------------------------------------------------------------------------------------------------------------------
# attention_schedule (attn.hip) through rgn_attention_plan_query: the remainder schedule per (Sq, Skv), as measured by
# tools/probes/attn_plan_sweep.py (profiles/r04_attn_plan_sweep_after_fix.txt)
# ---------------------------------------------------------------------------------------------------------------------------------
def aplan(Sq, Skv, H=24, ws=128 << 20):
    p = _lib.lib().rgn_attention_plan_query(Sq, Skv, H, ws)
    assert p >= 0, p
    return dict(pieces=p & 15, stream_k=bool(p & 16), waves8=bool(p & 32))


@pytest.mark.parametrize("Sq,Skv,want", [
    (1536, 8704, dict(pieces=1, stream_k=True, waves8=True)),        # FLUX region step at K_e 25 %: 144 items -> stream-K (171 vs 176 / 182 us)
    (1137, 8704, dict(pieces=2, stream_k=False, waves8=True)),       # K_e 15 %: 120 items x 2 equal pieces = one round (126 vs 149 us stream-K)
    (2048, 8704, dict(pieces=1, stream_k=False, waves8=True)),       # 192 items: the plain launch (194.5 vs 204 us stream-K) - round-4 fix
    (1536, 2560, dict(pieces=1, stream_k=False, waves8=True)),       # 144 items x 40 KV tiles: plain (56 us; the round-3 model split it: 70 us)
    (8704, 8704, dict(pieces=5, stream_k=False, waves8=True)),       # full step: 816 items = 3 rounds + 48 -> the remainder in 5 equal pieces
])
def test_attention_remainder_schedules(Sq, Skv, want):
    assert aplan(Sq, Skv) == want


def test_attention_plan_overrides_and_small_query_sets():
    assert not aplan(576, 8704)["waves8"]                            # 72 items of 256 rows < 96: 4-wave workgroups of 128 rows
    assert aplan(1536, 8704, ws=0) == dict(pieces=1, stream_k=False, waves8=True)     # no workspace: no partials
    with _lib.plan_override(attn_streamk=0):
        assert not aplan(1536, 8704)["stream_k"]
    with _lib.plan_override(attn_streamk=1):
        assert aplan(1137, 8704)["stream_k"]
    with _lib.plan_override(attn_split=0):
        assert aplan(1536, 8704) == dict(pieces=1, stream_k=False, waves8=True)
    with _lib.plan_override(attn_waves=4):
        assert not aplan(1536, 8704)["waves8"]
    assert _lib.lib().rgn_attention_plan_query(0, 8704, 24, 0) < 0


def test_full_step_ff_up_pair_splits_its_96_tile_remainder():
    # 1632 tiles = 6 whole rounds + 96: two K pieces for the remainder (522 -> 501 us measured; the 2.5 % acceptance threshold of round 4)
    assert plan((8192, 512), 12288, 3072) == dict(big=True, pieces=2, quarter=False)
    assert plan((8192, 512), 9216, 3072) == dict(big=True, pieces=1, quarter=False)          # Q/K/V pair: 4 rounds + 200, plain (364 vs 396 us)
