"""Launch-plan knobs for tests (rgn_plan_override, include/regione_hip.h): force a schedule so that a small problem reaches a
path it would not take by itself.  `force(...)` holds until the test ends (conftest resets every knob after each test);
`regione_amd._lib.plan_override(...)` is the `with`-scoped form."""
from regione_amd import _lib

# tile geometries of the GEMM: all three are live paths (128: small / ragged problems and quarter-tile remainders; 256c: the
# compiler-scheduled 8-wave kernel = the fallback for K < 128, operands >= 4 GiB and the fused-Q/K/V reduce pass; 256: the
# hand-scheduled 4-wave loop, the default from ~200 tiles up)
GEOMETRIES = {"128": dict(gemm_geometry=128), "256c": dict(gemm_geometry=256, gemm_asm=0), "256": dict(gemm_geometry=256, gemm_asm=-1)}


def force(**knobs):
    h = _lib.lib()
    for k, v in knobs.items():
        _lib.check(h.rgn_plan_override(k.encode(), int(v)), f"rgn_plan_override({k})")


def geometry(name: str):
    force(**GEOMETRIES[name])


def reset():
    _lib.lib().rgn_plan_override(None, 0)
