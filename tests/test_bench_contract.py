"""bench.py pieces that can be checked without a GPU: the PMC traffic file is only used for the kernel sources it was measured on
("make `traffic` refuse to load a PMC file whose kernel names/HEAD differ"), and the roofline object carries the fields
the contract names."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_stale_pmc_capture_is_refused(monkeypatch):
    b = _bench()
    files = [f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("pmc_traffic.json")]
    assert files, "no PMC capture committed"
    newest = json.load(open(os.path.join(ROOT, "profiles", sorted(files)[-1])))
    # a capture is accepted only for the sources it names ...
    monkeypatch.setattr(b, "kernel_source_sha", lambda: newest["kernel_source_sha"])
    got = b.pmc_traffic(newest["width"], newest["height"], newest["batch"])
    assert got is not None and got["hbm_bytes_per_call"] == newest["hbm_bytes_per_call"]
    # ... and for the workload it was taken on
    assert b.pmc_traffic(newest["width"] + 2, newest["height"], newest["batch"]) is None
    # ... and in the pyramid precision mode it was taken in: binary16 planes move half the bytes, an fp32 capture must not price them
    # (round 6 commits a capture of the binary16 mode as well: the other mode's query gets THAT file or nothing, never this one)
    other = b.pmc_traffic(newest["width"], newest["height"], newest["batch"], fp16=not newest.get("fp16", False))
    assert other is None or (bool(other.get("fp16", False)) != bool(newest.get("fp16", False)) and other["hbm_bytes_per_call"] != newest["hbm_bytes_per_call"])
    # any other kernel source hash: no traffic figure rather than a stale one
    monkeypatch.setattr(b, "kernel_source_sha", lambda: "0" * 16)
    assert b.pmc_traffic(newest["width"], newest["height"], newest["batch"]) is None


def test_roofline_object_fields():
    b = _bench()
    acc = {"nb_calls": 8, "nb_blur_launches": 40, "pyramid_ms": 14.0, "scan_ms": 6.0, "pyramid_algorithmic_bytes": 8 * 11.36e9, "scan_algorithmic_bytes": 8 * 3.15e9}
    own = 8 * 11.36e9 * 41.25 / 72.25 + 8 * 3.15e9 * 24.0 / 20.0
    r = b.roofline_from(acc, None, "label")
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["traffic"] is None
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # without a PMC capture: this build's own algorithmic minimum, never the reference-schedule pricing
    assert abs(r["achieved"] - own / 20.0e-3 / 1e9) < 1e-6 and r["algorithmic"]["frac"] == r["frac"]
    assert abs(r["survey_8d"]["achieved"] - (8 * 11.36e9 + 8 * 3.15e9) / 20.0e-3 / 1e9) < 1e-6
    assert r["launches_per_call"] == 6.0 and abs(r["algorithmic"]["bytes_per_launch"] - own / 48) < 1.0
    assert "pyramid_only" not in r
    # with one: frac IS the hardware figure (HBM bytes of the capture / the durations of this run), and traffic >= algorithmic
    pmc = {"hbm_bytes_per_call": 12.0e9, "_path": "profiles/x.json", "per_launch": [{"kernel": "k", "hbm_bytes": 1.0, "avg_us": 1.0}]}
    r2 = b.roofline_from(acc, pmc, "label")
    assert abs(r2["traffic"] - 12.0e9 / 6.0) < 1.0 and abs(r2["frac"] - 12.0e9 * 8 / 20.0e-3 / 1e9 / 8000.0) < 1e-9
    assert abs(r2["frac"] - r2["achieved"] / r2["peak"]) < 1e-12 and r2["algorithmic"]["frac"] == r["frac"]
    assert abs(r2["traffic_over_algorithmic"] - 12.0e9 * 8 / own) < 1e-9 and r2["per_launch"] == pmc["per_launch"]


def test_whole_pass_roofline_prices_every_octave():
    """round 6: `frac` covers every octave's scale-space launches + the scan; rounds 1-5's octave-0 figure moves to octave0_and_scan"""
    b = _bench()
    octs = [(1280, 960), (640, 480), (320, 240), (160, 120), (80, 60)]
    per_img = b.own_pyramid_bytes(octs, 640 * 480)
    p = [w * h for w, h in octs]
    assert abs(per_img - (41.25 * p[0] + 37 * (p[1] + p[2] + p[3]) + 36 * p[4])) < 1.0
    acc = {"nb_calls": 2, "nb_blur_launches": 10, "nb_blur_launches_all": 44, "pyramid_ms": 10.0, "pyramid_all_ms": 14.0, "scan_ms": 8.0,
           "pyramid_algorithmic_bytes": 2 * 512 * 72.25 * p[0], "scan_algorithmic_bytes": 2 * 512 * 20.0 * sum(p)}
    alg = 2 * 512 * (per_img + 24.0 * sum(p))
    r = b.roofline_from(acc, None, "label", octs, 640 * 480, 512)
    assert abs(r["achieved"] - alg / 22.0e-3 / 1e9) < 1e-3 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12 and r["traffic"] is None
    assert r["launches_per_call"] == 23.0 and "all octaves" in r["kernel"] and r["algorithmic"]["frac"] == r["frac"]
    o0 = r["octave0_and_scan"]
    assert abs(o0["achieved"] - 2 * 512 * (41.25 * p[0] + 24.0 * sum(p)) / 18.0e-3 / 1e9) < 1e-3 and o0["launches_per_call"] == 6.0
    pmc = {"hbm_bytes_per_call": 50.0e9, "_path": "profiles/x.json",
           "all_octaves": {"hbm_bytes_per_call": 58.0e9, "hbm_bytes_blur_per_call": 36.0e9, "launches": [{"kernel": "k", "hbm_bytes": 1.0}]}}
    r2 = b.roofline_from(acc, pmc, "label", octs, 640 * 480, 512)
    assert abs(r2["frac"] - 58.0e9 * 2 / 22.0e-3 / 1e9 / 8000.0) < 1e-9 and abs(r2["traffic_over_algorithmic"] - 58.0e9 * 2 / alg) < 1e-9
    assert abs(r2["octave0_and_scan"]["frac"] - 50.0e9 * 2 / 18.0e-3 / 1e9 / 8000.0) < 1e-9 and r2["all_launches"] == pmc["all_octaves"]["launches"]
    assert abs(r2["frac"] - r2["achieved"] / r2["peak"]) < 1e-12 and r2["algorithmic"]["frac"] == r["frac"]
