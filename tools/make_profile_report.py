#!/usr/bin/env python3
"""Turn a gpurun_out/<tag>/ capture (bench.json, trace/, pmc_fetch/, pmc_write/) into the committed files under
profiles/: <tag>_bench.json, <tag>_kernel_stats.csv, <tag>_kernel_summary.txt, <tag>_pmc_traffic.json.
usage: make_profile_report.py gpurun_out/r01 r01 <width> <height> <batch> [fp16]"""
import glob
import io
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


KERNEL = "k_blur_lean"
SCAN = "k_extrema_lean"


def pmc(dirpath, counter, gridx, kernel=KERNEL):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), dirpath, counter, kernel] + ([str(gridx)] if gridx else []),
                         capture_output=True, text=True)
    return json.loads(out.stdout)


BLUR_KERNELS = ("k_blur_", "k_input_blit", "k_downsample", "k_octave_chain")


def pmc_by_shape(dirpath, counter):
    """{(kernel name, grid_x, grid_y): (dispatches, KiB summed)} of every scale-space launch in a --pmc capture"""
    import sqlite3
    from collections import defaultdict
    out = defaultdict(lambda: [0, 0.0])
    for p in glob.glob(os.path.join(dirpath, "**", "*.db"), recursive=True):
        cur = sqlite3.connect(p).cursor()
        cols = [r[1] for r in cur.execute("pragma table_info('counters_collection')")]
        name_col = "kernel_name" if "kernel_name" in cols else "name"
        per = {}
        rows = list(cur.execute(f"select {name_col}, value, dispatch_id, grid_size_x, grid_size_y from counters_collection where counter_name = ?", (counter,)))
        # an instance TIMES a whole-batch blur launch on every candidate memory range before its first detection (vksift_instance.c:
        # place_pyramid_buffers): those dispatches are not part of any detection call — everything in front of the first seed launch
        # (k_blur_lean<N, 1 | 2, ...>: the fused copy-in + seed blur, or the input blit) is left out
        seeds = [disp for name, _, disp, _, _ in rows if re.search(r"k_blur_lean<\d+, [12],", name) or "k_input_blit" in name]
        first = min(seeds) if seeds else 0
        for name, value, disp, gx, gy in rows:
            if disp >= first and any(k in name for k in BLUR_KERNELS):
                key = (name, int(gx), int(gy))
                per.setdefault(disp, [key, 0.0])[1] += float(value)
        for key, v in per.values():
            out[key][0] += 1
            out[key][1] += v
    return {k: (c, v) for k, (c, v) in out.items()}


def kernel_source_sha():
    import hashlib
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(ROOT, "vulkansift_amd", "csrc", "hip", "*.hip"))):
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def main():
    src, tag = sys.argv[1], sys.argv[2]
    w, h, batch = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    fp16 = len(sys.argv) > 6 and sys.argv[6] == "fp16"   # a capture of the binary16 scale-space mode (bench.py --fp16): stamped, bench.py refuses it for fp32
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    if os.path.exists(os.path.join(src, "bench.json")):
        shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench.json"))
    have_trace = os.path.isdir(os.path.join(src, "trace"))
    csvs = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    for p in csvs:
        shutil.copy(p, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    if not csvs and have_trace:
        # this rocprofv3 writes a rocpd database instead of csv files: rebuild the --stats table from its kernel records
        import sqlite3
        from collections import defaultdict
        rows = defaultdict(list)
        for dbp in glob.glob(os.path.join(src, "trace", "**", "*.db"), recursive=True):
            for name, st, en in sqlite3.connect(dbp).execute("select name, start, end from kernels"):
                rows[name].append(en - st)
        tot = sum(sum(v) for v in rows.values()) or 1
        with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
            f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
            for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
                f.write('"%s",%d,%d,%.1f,%.2f,%d,%d\n' % (name, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / tot, min(v), max(v)))
    summ = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"), os.path.join(src, "trace"), KERNEL + ",k_blur_wide,k_blur_pair," + SCAN + ",k_descriptor,k_orientation<"],
                          capture_output=True, text=True).stdout if have_trace else ""
    if have_trace:
        open(os.path.join(dst, f"{tag}_kernel_summary.txt"), "w").write(summ)
    if os.path.isdir(os.path.join(src, "pmc_fetch")) and os.path.isdir(os.path.join(src, "pmc_write")):
        # octave-0 launches only (the launches bench.py's roofline is quoted on): k_blur_lean with grid.x = strips * 64 work-items
        # (128-column strips) and k_blur_pair (two scales per launch, 112 owned columns per strip)
        gridx = ((2 * w + 127) // 128) * 64
        gridx_pair = ((2 * w + 111) // 112) * 64
        f = pmc(os.path.join(src, "pmc_fetch"), "FETCH_SIZE", gridx)
        wr = pmc(os.path.join(src, "pmc_write"), "WRITE_SIZE", gridx)
        f.update(pmc(os.path.join(src, "pmc_fetch"), "FETCH_SIZE", gridx_pair, "k_blur_pair"))
        wr.update(pmc(os.path.join(src, "pmc_write"), "WRITE_SIZE", gridx_pair, "k_blur_pair"))
        # k_blur_wide (round 5): four texels per lane, 256-column strips
        gridx_wide = ((2 * w + 255) // 256) * 64
        f.update(pmc(os.path.join(src, "pmc_fetch"), "FETCH_SIZE", gridx_wide, "k_blur_wide"))
        wr.update(pmc(os.path.join(src, "pmc_write"), "WRITE_SIZE", gridx_wide, "k_blur_wide"))
        # k_blur_pair_wide (round 5): the two-scale launch with four texels per lane, 240 owned columns per strip
        gridx_pw = ((2 * w + 239) // 240) * 64
        f.update(pmc(os.path.join(src, "pmc_fetch"), "FETCH_SIZE", gridx_pw, "k_blur_pair_wide"))
        wr.update(pmc(os.path.join(src, "pmc_write"), "WRITE_SIZE", gridx_pw, "k_blur_pair_wide"))
        # the streaming extrema scan: ONE launch per detection over all octaves (flat grid) since round 3 — every dispatch of the kernel
        sf = pmc(os.path.join(src, "pmc_fetch"), "FETCH_SIZE", None, SCAN)
        sw_ = pmc(os.path.join(src, "pmc_write"), "WRITE_SIZE", None, SCAN)
        calls = sum(v["calls"] for v in f.values())
        fetch_kb = sum(v["sum"] for v in f.values())
        write_kb = sum(v["sum"] for v in wr.values())
        wcalls = sum(v["calls"] for v in wr.values())
        scalls = sum(v["calls"] for v in sf.values()) or 1
        swcalls = sum(v["calls"] for v in sw_.values()) or 1
        # FETCH_SIZE / WRITE_SIZE are in KiB. On gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced
        # stream (MI355X_MICROARCH.md §HBM): doubled before use. WRITE_SIZE is taken as is (uncalibrated).
        per_launch = (2.0 * fetch_kb / max(calls, 1) + write_kb / max(wcalls, 1)) * 1024.0
        scan_launch = (2.0 * sum(v["sum"] for v in sf.values()) / scalls + sum(v["sum"] for v in sw_.values()) / swcalls) * 1024.0
        blur_per_call = calls / scalls          # blur launches of octave 0 per detection call (= per scan launch)
        # per launch kind: HBM bytes (PMC) / average and minimum duration of the same launch shape in the kernel trace
        per_launch_rows = []
        durs, durs3 = {}, {}
        if have_trace:
            import sqlite3
            for dbp in glob.glob(os.path.join(src, "trace", "**", "*.db"), recursive=True):
                for name, st, en, gx, gy in sqlite3.connect(dbp).execute("select name, start, end, grid_x, grid_y from kernels"):
                    durs.setdefault((name, int(gx)), []).append(en - st)
                    durs3.setdefault((name, int(gx), int(gy)), []).append(en - st)
            for name in list(f.keys()) + list(sf.keys()):
                is_scan = name in sf
                fk = (sf if is_scan else f)[name]
                wk = (sw_ if is_scan else wr).get(name, {"avg": 0.0})
                gx = None if is_scan else (gridx_pw if "k_blur_pair_wide" in name else (gridx_pair if "k_blur_pair" in name else (gridx_wide if "k_blur_wide" in name else gridx)))
                dd = [v for (n, g), vs in durs.items() if n == name and (gx is None or g == gx) for v in vs]
                if not dd:
                    continue
                hbm = (2.0 * fk["avg"] + wk["avg"]) * 1024.0
                avg_us, min_us = sum(dd) / len(dd) / 1e3, min(dd) / 1e3
                per_launch_rows.append({"kernel": name.replace("void (anonymous namespace)::", "").split("(")[0], "hbm_bytes": hbm, "avg_us": avg_us, "min_us": min_us,
                                        "frac_of_8TBps": hbm / (avg_us * 1e-6) / 8e12, "frac_at_min_duration": hbm / (min_us * 1e-6) / 8e12, "launches_in_trace": len(dd)})
        # every scale-space launch of a detection call, all octaves (round 6: the whole pass is what bench.py prices): FETCH_SIZE x 2 + WRITE_SIZE
        # of every dispatch of the blur / blit / down-sampling kernels, grouped by launch shape
        allf, allw = pmc_by_shape(os.path.join(src, "pmc_fetch"), "FETCH_SIZE"), pmc_by_shape(os.path.join(src, "pmc_write"), "WRITE_SIZE")
        all_rows, all_bytes_per_call = [], 0.0
        for key in sorted(allf, key=lambda k: -allf[k][1]):
            name, gx, gy = key
            fc, fs = allf[key]
            wc, ws = allw.get(key, (0, 0.0))
            hbm = (2.0 * fs / max(fc, 1) + ws / max(wc, 1)) * 1024.0
            per_call = fc / scalls
            all_bytes_per_call += hbm * per_call
            dd = durs3.get(key, [])
            row = {"kernel": name.replace("void (anonymous namespace)::", "").split("(")[0], "grid_x": gx, "grid_y": gy, "launches_per_call": per_call, "hbm_bytes": hbm}
            if dd and per_call > 0:
                row.update({"avg_us": sum(dd) / len(dd) / 1e3, "min_us": min(dd) / 1e3, "frac_of_8TBps": hbm / (sum(dd) / len(dd) * 1e-9) / 8e12})
            all_rows.append(row)
        rec = {"fp16": fp16, "per_launch": per_launch_rows, "all_octaves": {"hbm_bytes_blur_per_call": all_bytes_per_call, "hbm_bytes_per_call": all_bytes_per_call + scan_launch,
                                                              "launches": all_rows}, "width": w, "height": h, "batch": batch, "kernel": KERNEL + " / k_blur_wide / k_blur_pair (octave-0 launches) + " + SCAN + " (one launch over all octaves)", "kernel_source_sha": kernel_source_sha(),
               "launches_fetch_pass": calls, "launches_write_pass": wcalls, "scan_launches": scalls,
               "FETCH_SIZE_KiB_sum": fetch_kb, "WRITE_SIZE_KiB_sum": write_kb, "fetch_correction": 2.0,
               "hbm_bytes_per_blur_launch": per_launch, "hbm_bytes_per_scan_launch": scan_launch,
               "hbm_bytes_per_call": per_launch * blur_per_call + scan_launch,
               "per_kernel_FETCH_SIZE": {**f, **sf}, "per_kernel_WRITE_SIZE": {**wr, **sw_}}
        json.dump(rec, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
        print("traffic per blur launch: %.1f MB, per scan launch: %.1f MB, per detection call: %.1f MB" % (per_launch / 1e6, scan_launch / 1e6, rec["hbm_bytes_per_call"] / 1e6))
    print(summ[:3000])


if __name__ == "__main__":
    main()
