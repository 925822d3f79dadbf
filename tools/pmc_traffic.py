#!/usr/bin/env python3
"""Per-launch HBM-side traffic of the MFMA kernels from two rocprofv3 PMC passes (rocpd sqlite output).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d out/rd -o rd -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-vanilla
    rocprofv3 --kernel-trace --pmc WRITE_SIZE                            -d out/wr -o wr -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-vanilla
    python tools/pmc_traffic.py out/rd/rd_results.db out/wr/wr_results.db > profiles/rNN_pmc_traffic.json

FETCH_SIZE itself segfaults rocprofv3 on this image, so the read side is rebuilt from its expression
((RDREQ - RDREQ_32B) * 64 + RDREQ_32B * 32 bytes) and doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-byte
requests are tallied at 64 B).  WRITE_SIZE is in KiB.  These are L2 -> fabric requests: Infinity-Cache hits are included."""
import json
import sqlite3
import sys


def table(c, prefix):
    return [r[0] for r in c.execute("select name from sqlite_master where type='table'") if r[0].startswith(prefix)][0]


def per_kernel(db):
    c = sqlite3.connect(db)
    ev, info, disp, sym = (table(c, "rocpd_pmc_event"), table(c, "rocpd_info_pmc"), table(c, "rocpd_kernel_dispatch"),
                           table(c, "rocpd_info_kernel_symbol"))
    cols = [r[1] for r in c.execute(f"pragma table_info({ev})")]
    key = "event_id" if "event_id" in cols else "dispatch_id"
    dkey = "event_id" if key == "event_id" else "id"
    q = (f"select s.kernel_name, p.name, count(*), sum(e.value) from {ev} e join {info} p on e.pmc_id = p.id "
         f"join {disp} d on e.{key} = d.{dkey} join {sym} s on d.kernel_id = s.id group by s.kernel_name, p.name")
    out = {}
    for name, counter, n, total in c.execute(q):
        # attention_asm_kernel (hand-scheduled loop) and attention_kernel (ragged KV lengths) count as one kernel family;
        # the small merge kernels (attention_combine*) are left out
        k = ("gemm_bf16_kernel" if ("gemm_bf16_kernel" in name or "gemm_reduce4w_kernel" in name) else
             "attention_kernel" if ("attention_kernel" in name or "attention_asm_kernel" in name) else None)
        if k:
            d = out.setdefault(k, {}).setdefault(counter, [0, 0.0])
            d[0] += n
            d[1] += total
    return out


def by_grid(db, top=14):
    """Read bytes (x2-corrected) per dispatch grouped by (kernel template, grid size): the grid identifies the projection (tiles of
    the launch), so the fabric traffic of ONE shape can be set against its operand bytes."""
    c = sqlite3.connect(db)
    ev, info, disp, sym = (table(c, "rocpd_pmc_event"), table(c, "rocpd_info_pmc"), table(c, "rocpd_kernel_dispatch"),
                           table(c, "rocpd_info_kernel_symbol"))
    dcols = [r[1] for r in c.execute(f"pragma table_info({disp})")]
    gx, wx = ("grid_size_x", "workgroup_size_x") if "grid_size_x" in dcols else (None, None)
    if gx is None:
        return []
    cols = [r[1] for r in c.execute(f"pragma table_info({ev})")]
    key = "event_id" if "event_id" in cols else "dispatch_id"
    dkey = "event_id" if key == "event_id" else "id"
    q = (f"select s.kernel_name, d.{gx} / d.{wx}, p.name, count(distinct d.id), sum(e.value) from {ev} e join {info} p on e.pmc_id = p.id "
         f"join {disp} d on e.{key} = d.{dkey} join {sym} s on d.kernel_id = s.id group by s.kernel_name, d.{gx} / d.{wx}, p.name")
    acc = {}
    for name, blocks, counter, n, total in c.execute(q):
        if "gemm" not in name and "attention" not in name:
            continue
        short = name.split("(")[0].replace("void rgn::", "")
        d = acc.setdefault((short, int(blocks)), {"n": n})
        d[counter] = total
    rows = []
    for (short, blocks), d in acc.items():
        if "TCC_EA0_RDREQ_sum" not in d:
            continue
        req, r32 = d["TCC_EA0_RDREQ_sum"], d.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        rows.append(dict(kernel=short, workgroups=blocks, dispatches=d["n"], read_bytes_x2_per_dispatch=2 * ((req - r32) * 64 + r32 * 32) / d["n"]))
    rows.sort(key=lambda r: -r["read_bytes_x2_per_dispatch"] * r["dispatches"])
    return rows[:top]


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("rd_db")
    ap.add_argument("wr_db")
    ap.add_argument("--edits", type=int, default=2, help="RegionE edits the profiled command ran (`bench.py --steps 1 --warmup 0 --no-5pct "
                                                         "--no-vanilla` = 1 timed + 1 characterising = 2)")
    ap.add_argument("--lat-db", default=None, help="optional third pass: --pmc TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum")
    args = ap.parse_args()
    rd, wr = per_kernel(args.rd_db), per_kernel(args.wr_db)
    res = {}
    for k in rd:
        n = rd[k]["TCC_EA0_RDREQ_sum"][0]
        req, req32 = rd[k]["TCC_EA0_RDREQ_sum"][1], rd[k]["TCC_EA0_RDREQ_32B_sum"][1]
        raw = ((req - req32) * 64 + req32 * 32) / n
        w = wr[k]["WRITE_SIZE"][1] * 1024 / wr[k]["WRITE_SIZE"][0]
        # per DISPATCH (a kernel launch; an op launch = whole rounds + remainder pieces + reduce / merge pass: several dispatches) and
        # per EDIT (what bench.py divides by its op launches per edit)
        res[k] = dict(dispatches=n, dispatches_per_edit=n / args.edits, read_bytes_raw_per_dispatch=raw, read_bytes_x2_per_dispatch=2 * raw,
                      write_bytes_per_dispatch=w, traffic_bytes_per_dispatch=2 * raw + w, traffic_bytes_per_edit=(2 * raw + w) * n / args.edits,
                      read_bytes_x2_per_edit=2 * raw * n / args.edits, write_bytes_per_edit=w * n / args.edits)
    if args.lat_db:
        # No Infinity-Cache (MALL) hit / miss or HBM-side (UMC / DF) counter is exposed by rocprofv3 on this image (blocks listed by
        # `rocprofv3 -L`: CPC CPF GRBM SPI SQ SQC TA TCA TCC TCP TD - profiles/r05_counter_blocks.txt).  The closest the TCC offers:
        # the average latency of an L2 -> fabric read (RDREQ_LEVEL / RDREQ, in L2 clocks) - a MALL hit returns sooner than an HBM read -
        # set against the same figure for a kernel that MUST stream from HBM (gemv_bf16_kernel over the 6.5 GB AdaLN table, 25 x the MALL);
        # RDREQ_DRAM counts requests ROUTED to local memory (vs GMI / IO), hits included: it equals RDREQ here.
        lat = {}
        c = sqlite3.connect(args.lat_db)
        ev, info, disp, sym = (table(c, "rocpd_pmc_event"), table(c, "rocpd_info_pmc"), table(c, "rocpd_kernel_dispatch"),
                               table(c, "rocpd_info_kernel_symbol"))
        cols = [r[1] for r in c.execute(f"pragma table_info({ev})")]
        key = "event_id" if "event_id" in cols else "dispatch_id"
        dkey = "event_id" if key == "event_id" else "id"
        q = (f"select s.kernel_name, p.name, sum(e.value) from {ev} e join {info} p on e.pmc_id = p.id "
             f"join {disp} d on e.{key} = d.{dkey} join {sym} s on d.kernel_id = s.id group by s.kernel_name, p.name")
        for name, counter, total in c.execute(q):
            fam = ("gemm_bf16_kernel" if ("gemm_bf16_kernel" in name or "gemm_reduce4w" in name) else
                   "attention_kernel" if ("attention_kernel" in name or "attention_asm_kernel" in name) else
                   "gemv_bf16_kernel (HBM-streaming yardstick)" if "gemv_bf16_kernel" in name else
                   "ln_modulate_kernel" if "ln_modulate" in name else None)
            if fam:
                lat.setdefault(fam, {})[counter] = lat.setdefault(fam, {}).get(counter, 0.0) + total
        for fam, d in lat.items():
            if d.get("TCC_EA0_RDREQ_sum"):
                d["avg_fabric_read_latency_l2_clocks"] = d.get("TCC_EA0_RDREQ_LEVEL_sum", 0.0) / d["TCC_EA0_RDREQ_sum"]
                d["share_routed_to_local_dram"] = d.get("TCC_EA0_RDREQ_DRAM_sum", 0.0) / d["TCC_EA0_RDREQ_sum"]
        for k in res:
            if k in lat:
                res[k]["hbm_side"] = dict(lat[k], yardstick=lat.get("gemv_bf16_kernel (HBM-streaming yardstick)"),
                                          note="no MALL / UMC counter in rocprofv3 on this image; latency of L2->fabric reads vs an HBM-streaming kernel")
        res["fabric_read_latency_by_family"] = lat
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_hash
    try:
        res["reads_by_kernel_and_grid"] = by_grid(args.rd_db)
    except sqlite3.Error as e:                # schema differences between rocprofv3 versions: the totals above do not depend on it
        res["reads_by_kernel_and_grid"] = f"unavailable: {e}"
    res["csrc_sha16"] = csrc_hash()          # bench.py quotes this file only for a build of the same kernel sources
    res["edits"] = args.edits
    res["note"] = ("rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum (own pass) and --pmc WRITE_SIZE (own pass) over "
         "`bench.py --steps 1 --warmup 0 --no-5pct --no-vanilla`; read bytes = ((RDREQ-RDREQ_32B)*64 + RDREQ_32B*32), doubled per MI355X_MICROARCH.md "
         "(gfx950 tallies 128-B requests at 64 B); WRITE_SIZE in KiB.  L2->fabric requests: Infinity-Cache hits are included.")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
