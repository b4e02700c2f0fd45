#!/usr/bin/env python3
"""Collect a set of SQ counters for one kernel with several rocprofv3 --pmc passes (few counters fit per pass) and print
per-dispatch averages. Run ON the GPU box:  python tools/pmc_kernel.py <kernel-substring> -- <command ...>"""
import glob, json, os, sqlite3, subprocess, sys, tempfile
GROUPS = [["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU"],
          ["SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_MFMA"],
          ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_ANY"],
          ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_INSTS_BRANCH"],
          ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INST_LEVEL_LDS"]]
if os.environ.get("PMC_GROUPS"):   # e.g. PMC_GROUPS="SQ_INSTS_VALU,SQ_BUSY_CU_CYCLES;TA_TA_BUSY_sum,GRBM_GUI_ACTIVE"
    GROUPS = [g.split(",") for g in os.environ["PMC_GROUPS"].split(";") if g]
name = sys.argv[1]
cmd = sys.argv[sys.argv.index("--") + 1:]
res = {}
for g in GROUPS:
    d = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
    try:   # a group the hardware cannot schedule together can stall the profiler: bounded, and the other groups still report
        subprocess.run(["rocprofv3", "--pmc", *g, "-d", d, "--"] + cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp",
                       env=dict(os.environ, TMPDIR="/tmp"), timeout=float(os.environ.get("PMC_PASS_TIMEOUT", "180")))
    except subprocess.TimeoutExpired:
        print("timeout:", g, file=sys.stderr, flush=True)
        continue
    for p in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        db = sqlite3.connect(p)
        acc = {}
        for kn, cn, v, disp in db.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
            if name in kn:
                acc.setdefault(cn, {}).setdefault(disp, 0.0)
                acc[cn][disp] += float(v)
        for cn, dd in acc.items():
            res[cn] = sum(dd.values()) / max(len(dd), 1)
    print("done:", g, {k: res.get(k) for k in g}, file=sys.stderr, flush=True)
print(json.dumps(res, indent=1))
