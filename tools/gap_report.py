"""Largest idle gaps between kernels in the steady state of a rocprofv3 --kernel-trace --memory-copy-trace database, with the kernels on
both sides and the copies in flight during the gap. usage: gap_report.py <dir> [min_gap_us]"""
import glob, sqlite3, sys
rows, cps = [], []
for p in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    db = sqlite3.connect(p)
    rows += list(db.execute("select name,start,end from kernels"))
    cps += list(db.execute("select name,start,end,size from memory_copies"))
rows.sort(key=lambda r: r[1])
ming = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 100e3
lo = rows[len(rows) // 3][1]
t0 = lo
def short(n): return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:34]
end = None
tot = 0
for i, (n, s, e) in enumerate(rows):
    if s < lo:
        end = max(end or e, e); continue
    if end is not None and s - end > ming:
        inflight = [(c[0].replace("MEMORY_COPY_", ""), (c[1] - t0) / 1e6, (c[2] - c[1]) / 1e3, c[3]) for c in cps if c[1] < s and c[2] > end]
        print(f"gap {(s - end) / 1e3:8.1f} us at {(end - t0) / 1e6:8.2f} ms  after {short(rows[i - 1][0])}  before {short(n)}  copies in flight: {[(a, round(b, 2), round(c), d) for a, b, c, d in inflight][:4]}")
        tot += s - end
    end = max(end or e, e)
print("total idle in listed gaps ms", tot / 1e6, "window ms", (rows[-1][2] - lo) / 1e6)
