#!/usr/bin/env python3
"""Per-kernel averages of the counters of ONE rocprofv3 --pmc pass (rocpd sqlite output), MFMA kernels only.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES \
              SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d out/mfma -o m -- \
              python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-vanilla
    python tools/pmc_summary.py out/mfma/m_results.db > profiles/rNN_pmc_mfma.json

Derived (gfx94x formula, ROCm 7.2 ships no gfx950 section): MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 256 CUs * 4 SIMDs);
LDS bank-conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import table  # noqa: E402

CUS, SIMDS = 256, 4


def main():
    c = sqlite3.connect(sys.argv[1])
    ev, info, disp, sym = (table(c, "rocpd_pmc_event"), table(c, "rocpd_info_pmc"), table(c, "rocpd_kernel_dispatch"),
                           table(c, "rocpd_info_kernel_symbol"))
    cols = [r[1] for r in c.execute(f"pragma table_info({ev})")]
    key = "event_id" if "event_id" in cols else "dispatch_id"
    dkey = "event_id" if key == "event_id" else "id"
    q = (f"select s.kernel_name, e.{key}, p.name, count(*), sum(e.value) from {ev} e join {info} p on e.pmc_id = p.id "
         f"join {disp} d on e.{key} = d.{dkey} join {sym} s on d.kernel_id = s.id group by s.kernel_name, e.{key}, p.name")
    # one row per (dispatch, counter): the counter's instances (XCC / SE dimensions) summed, and how many there were
    agg = {}
    for name, _, counter, rows, total in c.execute(q):
        if "gemm_bf16_kernel" in name or "gemm_reduce4w_kernel" in name or "attention_kernel" in name or "attention_asm_kernel" in name:
            short = name.split("(")[0].replace("void rgn::", "")
            for k in (short, "ALL " + ("gemm_bf16_kernel" if "gemm" in name else "attention_kernel")):
                d = agg.setdefault(k, {}).setdefault(counter, [0, 0.0, 0])
                d[0] += 1
                d[1] += total
                d[2] += rows
    out = {}
    for k, cs in sorted(agg.items()):
        n = max(v[0] for v in cs.values())
        row = {"dispatches": n, "sum_over_instances_per_launch": {c_: v[1] / v[0] for c_, v in cs.items()},
               "instances": {c_: v[2] / v[0] for c_, v in cs.items()}}
        p, inst = row["sum_over_instances_per_launch"], row["instances"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in p and p.get("GRBM_GUI_ACTIVE"):
            active = p["GRBM_GUI_ACTIVE"] / inst["GRBM_GUI_ACTIVE"]            # cycles the launch was resident (mean instance)
            row["mfma_util"] = p["SQ_VALU_MFMA_BUSY_CYCLES"] / (active * CUS * SIMDS)
        if p.get("SQ_LDS_IDX_ACTIVE"):
            row["lds_bank_conflict_share"] = p.get("SQ_LDS_BANK_CONFLICT", 0.0) / p["SQ_LDS_IDX_ACTIVE"]
        out[k] = row
    out["note"] = ("one rocprofv3 --kernel-trace --pmc pass over `bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-vanilla`; "
                   "mfma_util = sum over instances of SQ_VALU_MFMA_BUSY_CYCLES / (mean GRBM_GUI_ACTIVE * 256 CUs * 4 SIMDs) (gfx94x derived-metric formula); kernels are "
                   "slowed by the counter collection, ratios are what to read")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_hash
    out["csrc_sha16"] = csrc_hash()           # bench.py quotes mfma_busy only for a build of the same kernel sources
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
