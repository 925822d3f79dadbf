#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel launches / total / avg / min / max.

    python tools/rocpd_summary.py gpurun_out/prof1/r01_results.db > profiles/r01_kernel_stats.txt
"""
import os
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    namecol = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else scols[1])
    q = (f"select s.{namecol}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
         f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.{namecol} order by 3 desc")
    rows = list(c.execute(q))
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace summary of {os.path.basename(path)}" + (f"  [{sys.argv[2]}]" if len(sys.argv) > 2 else ""))
    print(f"# {'kernel':<70} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}")
    for name, n, tot, mn, mx in rows:
        nm = name if len(name) <= 70 else name[:67] + "..."
        print(f"{nm:<72} {n:>7} {tot / 1e6:>10.3f} {tot / n / 1e3:>10.2f} {mn / 1e3:>9.2f} {mx / 1e3:>9.2f} {100.0 * tot / total:>6.2f}")
    print(f"# total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    # kernel families (all template instantiations): bench.py times one ops.gemm / ops.attention CALL per "launch" - a call
    # whose remainder is split runs 2-3 dispatches of the family - so compare family total per edit, not per-dispatch averages
    fam = {}
    for name, n, tot, mn, mx in rows:
        for key in ("gemm_bf16_kernel", "attention"):
            if key in name:
                f = fam.setdefault(key, [0, 0])
                f[0] += n
                f[1] += tot
    for key, (n, tot) in fam.items():
        print(f"# family {key:<20} {n:>7} dispatches {tot / 1e6:>10.3f} ms  ({100.0 * tot / total:.2f} % of kernel time)")


if __name__ == "__main__":
    main(sys.argv[1])
