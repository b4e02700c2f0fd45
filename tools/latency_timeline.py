"""Timeline of ONE single-image detection + download from a rocprofv3 --kernel-trace --memory-copy-trace database of tools/bench_latency.py.
usage: latency_timeline.py <dir>"""
import glob, sqlite3, sys
rows, cps = [], []
for p in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    db = sqlite3.connect(p)
    rows += [("K", n, s, e, 0) for n, s, e in db.execute("select name,start,end from kernels")]
    rows += [("C", n, s, e, sz) for n, s, e, sz in db.execute("select name,start,end,size from memory_copies")]
rows.sort(key=lambda r: r[2])
h2d = [i for i, r in enumerate(rows) if r[0] == "K" and "k_blur_lean<5, 1" in r[1]]   # the seed launch opens a detection
a, b = h2d[len(h2d) // 2], h2d[len(h2d) // 2 + 1]
t0 = rows[a][2]
prev_end = t0
busy = 0
for kind, n, s, e, sz in rows[a:b]:
    short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:6.1f} gap {(s - prev_end) / 1e3:6.1f}  {kind} {short} {sz if sz else ''}")
    prev_end = max(prev_end, e); busy += e - s
print("period us", (rows[b][2] - t0) / 1e3, "busy us", busy / 1e3, "items", b - a)
