"""From a rocprofv3 --kernel-trace --memory-copy-trace rocpd database: wall span, union of kernel busy time, per-kernel totals and copy totals
in the second half of the run (steady state). usage: busy_summary.py <dir>"""
import glob, os, sqlite3, sys
from collections import defaultdict
rows, copies = [], []
for p in glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True):
    db = sqlite3.connect(p)
    rows += list(db.execute("select name, start, end from kernels"))
    try:
        copies += list(db.execute("select name, start, end, size from memory_copies"))
    except Exception as e:
        print("no copy table:", e)
rows.sort(key=lambda r: r[1])
t_lo = rows[len(rows) // 2][1]; t_hi = rows[-1][2]
sel = [r for r in rows if r[1] >= t_lo]
iv = sorted((r[1], r[2]) for r in sel)
busy, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"window {(t_hi - t_lo) / 1e6:.2f} ms, kernel-busy union {busy / 1e6:.2f} ms, sum of kernel durations {sum(r[2] - r[1] for r in sel) / 1e6:.2f} ms")
by = defaultdict(float)
for n, s, e in sel: by[n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]] += (e - s) / 1e6
for n, t in sorted(by.items(), key=lambda kv: -kv[1])[:14]: print(f"  {t:8.2f} ms  {n}")
cs = [c for c in copies if c[1] >= t_lo]
byc = defaultdict(lambda: [0.0, 0, 0])
for n, s, e, sz in cs:
    byc[n][0] += (e - s) / 1e6; byc[n][1] += sz or 0; byc[n][2] += 1
for n, (t, sz, k) in byc.items(): print(f"  copies {n}: {k} x, {t:.2f} ms, {sz / 1e6:.1f} MB")
