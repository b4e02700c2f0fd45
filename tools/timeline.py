#!/usr/bin/env python3
"""Print the kernel timeline (per stream) of one steady-state step from a rocprofv3 --kernel-trace rocpd database.
usage: timeline.py <dir> [t0_ms] [t1_ms]   (times relative to the first kernel; default: a 12 ms window in the middle)"""
import glob, os, sqlite3, sys
d = sys.argv[1]
rows = []
for p in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
    db = sqlite3.connect(p)
    rows += list(db.execute("select name, start, end, stream_id, queue_id, grid_x, grid_y, grid_z from kernels"))
rows.sort(key=lambda r: r[1])
t00 = rows[0][1]
mid = (rows[len(rows) // 2][1] - t00) / 1e6
t0 = float(sys.argv[2]) if len(sys.argv) > 2 else mid
t1 = float(sys.argv[3]) if len(sys.argv) > 3 else t0 + 12.0
for name, s, e, stream, queue, gx, gy, gz in rows:
    a, b = (s - t00) / 1e6, (e - t00) / 1e6
    if b < t0 or a > t1:
        continue
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print(f"{a:9.3f} {b:9.3f} {b - a:7.3f}  st={stream:<3} q={queue:<3} {short:32s} grid=({gx},{gy},{gz})")
