"""Per-kernel launch count and mean duration in the steady state (second half) of two rocprofv3 --kernel-trace databases, side by side:
which kernels run longer in the second protocol, and by how much per 512-frame detection. usage: kernel_means.py <dirA> <dirB> [launches_per_detection_key]"""
import glob, os, sqlite3, sys
from collections import defaultdict
def load(d):
    rows = []
    for p in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        rows += list(sqlite3.connect(p).execute("select name, start, end from kernels"))
    rows.sort(key=lambda r: r[1])
    rows = rows[len(rows) // 2:]
    by = defaultdict(list)
    for n, s, e in rows:
        by[n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]].append((e - s) / 1e3)
    return by
A, B = load(sys.argv[1]), load(sys.argv[2])
key = sys.argv[3] if len(sys.argv) > 3 else "k_descriptor<2, false, false>"
na, nb = max(1, len(A.get(key, []))), max(1, len(B.get(key, [])))
print(f"detections in window: A {na}  B {nb} (launches of {key})")
tot = 0.0
for n in sorted(set(A) | set(B), key=lambda n: -(sum(B.get(n, [])) / nb - sum(A.get(n, [])) / na)):
    a, b = sum(A.get(n, [])) / na, sum(B.get(n, [])) / nb
    if abs(b - a) > 20 or a > 300:
        print(f"{n[:46]:46s} A {len(A.get(n, [])) / na:5.1f} x {a:9.1f} us/det   B {len(B.get(n, [])) / nb:5.1f} x {b:9.1f} us/det   delta {b - a:+8.1f}")
    tot += b - a
print(f"sum of kernel durations per detection: delta {tot:+.1f} us")
