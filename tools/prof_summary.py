#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace output directory (csv or rocpd sqlite) per kernel / per grid.
usage: prof_summary.py <dir> [name-filter[,name-filter...]]   (per-grid breakdown of the kernels matching any of the filters)"""
import csv
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def rows_from_db(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    for r in cur.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count from kernels"):
        yield {"name": r[0], "dur": r[2] - r[1], "grid": (r[3], r[4], r[5]), "wg": r[6], "lds": r[7], "vgpr": r[8]}


def rows_from_csv(path):
    for r in csv.DictReader(open(path)):
        yield {"name": r["Kernel_Name"], "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
               "grid": (int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])), "wg": int(r["Workgroup_Size_X"]),
               "lds": int(r.get("LDS_Block_Size", 0) or 0), "vgpr": int(r.get("VGPR_Count", 0) or 0)}


def main():
    d = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else None
    rows = []
    for p in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        rows += list(rows_from_db(p))
    for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(rows_from_csv(p))
    by = defaultdict(list)
    for r in rows:
        by[r["name"]].append(r)
    tot = sum(r["dur"] for r in rows)
    print(f"{'kernel':72s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'%':>6s}")
    for name, rs in sorted(by.items(), key=lambda kv: -sum(r['dur'] for r in kv[1])):
        t = sum(r["dur"] for r in rs)
        print(f"{name[:72]:72s} {len(rs):7d} {t/1e6:10.3f} {t/len(rs)/1e3:9.2f} {100*t/tot:6.1f}")
    if flt:
        flts = [f for f in flt.split(",") if f]
        print(f"\nper-grid breakdown of kernels matching {flts}:")
        g = defaultdict(list)
        for r in rows:
            if any(f in r["name"] for f in flts):
                g[(r["name"][:60], r["grid"], r["lds"], r["vgpr"])].append(r["dur"])
        for k, v in sorted(g.items(), key=lambda kv: (-kv[0][1][0] * kv[0][1][1], kv[0][0])):
            print(f"  {k[0]:60s} grid={k[1]} lds={k[2]} vgpr={k[3]} calls={len(v)} avg_us={sum(v)/len(v)/1e3:.2f} min_us={min(v)/1e3:.2f}")


if __name__ == "__main__":
    main()
