"""Consistency of what the repo SAYS with what it ships (no GPU): the counter files the bench line quotes belong to the committed
kernel sources, every environment switch the code reads is documented, DESIGN.md keeps the size the round-3 review asked for."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read(*parts):
    with open(os.path.join(ROOT, *parts), encoding="utf-8") as f:
        return f.read()


def test_committed_counter_files_carry_the_stamp_of_the_committed_kernel_sources():
    """bench.py quotes roofline.traffic / mfma_busy only from PMC summaries whose csrc_sha16 equals the hash of the kernel
    sources it runs; a kernel edit without a new measurement pass would silently drop both fields from the driver's line."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    stamp = bench.csrc_hash()
    import pytest
    if not all(os.path.exists(os.path.join(ROOT, n)) for n in (bench.PMC_TRAFFIC_FILE, bench.PMC_MFMA_FILE)):
        pytest.skip("no counter pass of this round committed yet (tools/probes/measure_counters.sh)")
    for name in (bench.PMC_TRAFFIC_FILE, bench.PMC_MFMA_FILE):
        assert json.loads(_read(name))["csrc_sha16"] == stamp, (name, stamp)
    # and a file with another stamp is NOT quoted
    assert bench.load_stamped(bench.PMC_TRAFFIC_FILE, stamp="0" * 16) == {}
    assert bench.load_stamped(bench.PMC_TRAFFIC_FILE).get("csrc_sha16") == stamp


def test_the_shipped_library_reads_one_environment_variable_and_every_switch_is_documented():
    """no per-launch getenv, <= 6 documented switches, A/B scaffolding out of the product."""
    n_getenv = 0
    for fn in os.listdir(os.path.join(ROOT, "regione_amd", "csrc")):
        if fn.endswith((".hip", ".h", ".inc", ".cpp")):
            n_getenv += len(re.findall(r"\bgetenv\(", _read("regione_amd", "csrc", fn)))
    assert n_getenv == 1, n_getenv                                  # RGN_PLAN_OVERRIDE, parsed once (region.hip: plan_override())
    pat = re.compile(r"(?:getenv\(\s*|environ\.get\(\s*|environ\[\s*)\"(RGN_[A-Z0-9_]+)\"")
    used = set()
    for base, _, files in os.walk(os.path.join(ROOT, "regione_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                used |= set(pat.findall(_read(base, fn)))
    used |= set(pat.findall(_read("bench.py")))
    assert used == {"RGN_LIB", "RGN_TORCH_OPS", "RGN_BATCH_BRANCHES", "RGN_BRANCH_STREAMS", "RGN_PLAN_OVERRIDE"}, sorted(used)
    doc = _read("INTEGRATION.md")
    assert all(v in doc for v in used)
    # every launch-plan knob the library knows is in the table too
    from regione_amd import _lib
    assert all(f"`{k}`" in doc for k in _lib.PLAN_KEYS)


def test_design_md_stays_under_40_kib():
    assert len(_read("DESIGN.md").encode("utf-8")) <= 40 * 1024
