#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counters from the rocpd sqlite output.
usage: pmc_summary.py <dir> <COUNTER> [name-filter] [grid_size_x]  -> prints JSON {kernel: {calls, sum, avg}}
(grid_size_x, in work-items, restricts the sums to one launch shape, e.g. the octave-0 launches of a blur kernel)"""
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict


def main():
    d, counter = sys.argv[1], sys.argv[2]
    flt = sys.argv[3] if len(sys.argv) > 3 else ""
    gridx = int(sys.argv[4]) if len(sys.argv) > 4 else None
    out = defaultdict(lambda: [0, 0.0])
    for p in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        db = sqlite3.connect(p)
        cur = db.cursor()
        cols = [r[1] for r in cur.execute("pragma table_info('counters_collection')")]
        # expected columns: ... kernel name, counter_name, value
        name_col = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else None)
        q = f"select {name_col}, counter_name, value, dispatch_id, grid_size_x from counters_collection where counter_name = ?"
        per_dispatch = defaultdict(float)
        names = {}
        for name, cname, value, disp, gx in cur.execute(q, (counter,)):
            if gridx is not None and int(gx) != gridx:
                continue
            per_dispatch[disp] += float(value)
            names[disp] = name
        for disp, v in per_dispatch.items():
            n = names[disp]
            if flt in n:
                out[n][0] += 1
                out[n][1] += v
    res = {k: {"calls": c, "sum": s, "avg": s / max(c, 1)} for k, (c, s) in out.items()}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
