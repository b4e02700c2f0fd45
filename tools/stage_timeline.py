"""Per-detection stage boundaries from a rocprofv3 --kernel-trace database of a batched run (resident bench or pipelined leg):
for every k_descriptor launch D_i: when the scale-space launches between D_{i-1} and D_i started / ended, when the scan, the
descriptor and the matching ran — all relative to the end of the previous descriptor launch. usage: stage_timeline.py <dir>"""
import glob, os, sqlite3, sys
rows = []
for p in glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True):
    rows += list(sqlite3.connect(p).execute("select name, start, end from kernels"))
rows.sort(key=lambda r: r[1])
def short(n): return n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
rows = [(short(n), s, e) for n, s, e in rows]
desc = [r for r in rows if r[0].startswith("k_descriptor")]
print("   period | blur first..last (busy) | scan start..end | ori+desc start..end | match first..last (busy) | pack busy")
for i in range(max(1, len(desc) - 6), len(desc)):
    t0, t1 = desc[i - 1][2], desc[i][2]
    win = [r for r in rows if r[1] >= t0 - 12e6 and r[1] < t1]
    blur = [r for r in win if r[0].startswith("k_blur") and r[1] >= desc[i - 1][1] - 2e6 and r[2] <= desc[i][1]]
    scan = [r for r in win if r[0].startswith("k_extrema") and r[1] > t0 - 1e6]
    ori = [r for r in win if r[0].startswith("k_orientation<") and r[1] > t0 - 1e6]
    mt = [r for r in win if r[0].startswith("k_match") and r[1] >= t0]
    pk = [r for r in win if r[0].startswith("k_pack") and r[1] >= t0]
    f = lambda t: (t - t0) / 1e6
    b = lambda L: sum(e - s for _, s, e in L) / 1e6
    print(f"{(t1 - t0) / 1e6:9.2f} | {f(blur[0][1]):6.2f}..{f(blur[-1][2]):6.2f} ({b(blur):5.2f}) | {f(scan[-1][1]):6.2f}..{f(scan[-1][2]):6.2f} | "
          f"{f(ori[-1][1]):6.2f}..{f(desc[i][2]):6.2f} | " + (f"{f(mt[0][1]):6.2f}..{f(mt[-1][2]):6.2f} ({b(mt):5.2f})" if mt else "   -   ") + f" | {b(pk):5.2f}")
