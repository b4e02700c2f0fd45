#!/usr/bin/env python3
"""SQ counters of the matcher kernels -> profiles/<tag>_sq_counters_match.json (run on the GPU box).
usage: capture_match_counters.py <tag>"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
runs = [
    ("k_match_scan32,k_match_fix,k_match_redo_rows", [sys.executable, os.path.join(ROOT, "tools", "match_time.py"), "50000", "3"], {}, "50000 x 50000, single pair (tools/match_time.py): cell scan + exact finish + replay"),
    ("k_match_mfma<,k_match_merge", [sys.executable, os.path.join(ROOT, "tools", "match_time.py"), "50000", "3"], {"VKSIFT_MATCH_SCAN": "0"}, "50000 x 50000 with VKSIFT_MATCH_SCAN=0: the stream-decomposed pruning kernel of rounds 2-3 (16x16x64 MFMA)"),
    ("k_match_pk", [sys.executable, os.path.join(ROOT, "tools", "match_time.py"), "13000", "3"], {}, "13000 x 13000, single pair: packed-key kernel <4,128> with 4 pieces of B"),
    ("k_match_pk", [sys.executable, os.path.join(ROOT, "tools", "batch_match_time.py"), "512", "3", "match-only"], {}, "512 self-matches of 1913 x 1913 (tools/batch_match_time.py): packed-key kernel <8,128>, 256 pairs per launch"),
]
out = {"note": "SQ counters per dispatch (tools/pmc_kernel.py: five rocprofv3 --pmc passes of four counters each). SQ_WAVE_CYCLES / SQ_WAIT_* / "
               "SQ_ACTIVE_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles, SQ_BUSY_CYCLES sums 32 shader engines."}
for names, cmd, env, what in runs:
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_kernel.py"), names, "--"] + cmd, capture_output=True, text=True,
                       env=dict(os.environ, PMC_PASS_TIMEOUT="200", **env))
    try:
        d = json.loads(r.stdout[r.stdout.index("{"):])
    except Exception as e:  # noqa: BLE001
        out[what] = {"error": repr(e), "stderr": r.stderr[-500:]}
        continue
    if "," not in names:
        d = {names: d}
    for k, v in d.items():
        if v.get("SQ_INSTS_MFMA"):
            m = v["SQ_INSTS_MFMA"]
            v["derived"] = {"valu_per_mfma": v["SQ_INSTS_VALU"] / m, "salu_per_mfma": v["SQ_INSTS_SALU"] / m, "lds_per_mfma": v["SQ_INSTS_LDS"] / m,
                            "mfma_pipe_busy_frac": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["SQ_BUSY_CYCLES"] / 32 * 1024),
                            "wave_wait_frac": v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], "wave_issue_stall_frac": v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"],
                            "lds_bank_conflict_frac": v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1)}
    out[what] = d
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_sq_counters_match.json"), "w"), indent=1)
for what, d in out.items():
    if isinstance(d, dict):
        for k, v in d.items():
            if isinstance(v, dict) and "derived" in v:
                print(what[:60], "|", k, {a: round(b, 3) for a, b in v["derived"].items()})
