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
    for name in ("r04_pmc_traffic.json", "r04_pmc_mfma.json"):
        assert json.loads(_read("profiles", name))["csrc_sha16"] == stamp, (name, stamp)
    assert stamp in _read("profiles", "README_r04.md")


def test_every_environment_switch_the_code_reads_is_documented():
    pat = re.compile(r"(?:getenv\(\s*|environ\.get\(\s*|environ\[\s*)\"(RGN_[A-Z0-9_]+)\"")
    used = set()
    for base, _, files in os.walk(os.path.join(ROOT, "regione_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                used |= set(pat.findall(_read(base, fn)))
    used |= set(pat.findall(_read("bench.py")))
    assert len(used) >= 20
    doc = _read("INTEGRATION.md")
    missing = sorted(v for v in used if v not in doc)
    # RGN_GEMM_DBG exists only in -DRGN_TIMING_PROBES builds and is described as such
    assert missing in ([], ["RGN_GEMM_DBG"]), missing


def test_design_md_stays_under_40_kib():
    assert len(_read("DESIGN.md").encode("utf-8")) <= 40 * 1024
