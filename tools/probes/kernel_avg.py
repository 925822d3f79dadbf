#!/usr/bin/env python3
"""Average duration per kernel name from a rocprofv3 rocpd database (`rocprofv3 --kernel-trace -d DIR -o NAME -- cmd`):
    python tools/probes/kernel_avg.py DIR/**/NAME_results.db [substr ...]"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    namecol = "display_name" if "display_name" in scols else "kernel_name"
    rows = c.execute(f"select s.{namecol}, count(*), avg(d.end - d.start), min(d.end - d.start), sum(d.end - d.start) from {kd} d "
                     f"join {ks} s on d.kernel_id = s.id group by s.{namecol} order by 5 desc")
    want = sys.argv[2:]
    for name, n, avg, mn, tot in rows:
        short = name.split("(")[0].replace("void rgn::", "").replace("rgn::", "")[:80]
        if not want or any(w in short for w in want):
            print(f"{short:<82} {n:>6} calls  avg {avg / 1e3:9.1f} us  min {mn / 1e3:9.1f} us  total {tot / 1e6:9.2f} ms")


if __name__ == "__main__":
    main()
