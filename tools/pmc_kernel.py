#!/usr/bin/env python3
"""Collect a set of SQ counters for one kernel with several rocprofv3 --pmc passes (few counters fit per pass) and print
per-dispatch averages. Run ON the GPU box:  python tools/pmc_kernel.py <kernel-substring>[,<substring>...] -- <command ...>
With several substrings the output is {substring: {counter: average per dispatch}} from the SAME passes. A substring may carry a
launch-shape filter "name@<grid_size_x>" (work-items), e.g. "k_blur_lean<5, 1@40960" = the octave-0 seed launch only."""
import glob, json, os, sqlite3, subprocess, sys, tempfile
GROUPS = [["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU"],
          ["SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_MFMA"],
          ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_ANY"],
          ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_INSTS_BRANCH"],
          ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INST_LEVEL_LDS"]]
if os.environ.get("PMC_GROUPS"):   # e.g. PMC_GROUPS="SQ_INSTS_VALU,SQ_BUSY_CU_CYCLES;TA_TA_BUSY_sum,GRBM_GUI_ACTIVE"
    GROUPS = [g.split(",") for g in os.environ["PMC_GROUPS"].split(";") if g]
names = [n for n in sys.argv[1].split(",") if n]
# a template argument list contains commas: "k_blur_lean<5, 1" arrives split -> glue pieces that start with a blank back together
glued = []
for n in names:
    if glued and n.startswith(" "):
        glued[-1] += "," + n
    else:
        glued.append(n)
names = glued
cmd = sys.argv[sys.argv.index("--") + 1:]
res = {n: {} for n in names}
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
        acc = {n: {} for n in names}
        for kn, cn, v, disp, gx in db.execute("select kernel_name, counter_name, value, dispatch_id, grid_size_x from counters_collection"):
            for n in names:
                sub, _, shape = n.partition("@")
                if sub in kn and (not shape or int(gx) == int(shape)):
                    acc[n].setdefault(cn, {}).setdefault(disp, 0.0)
                    acc[n][cn][disp] += float(v)
        for n in names:
            for cn, dd in acc[n].items():
                res[n][cn] = sum(dd.values()) / max(len(dd), 1)
                res[n]["dispatches"] = len(dd)
    print("done:", g, file=sys.stderr, flush=True)
print(json.dumps(res if len(names) > 1 else res[names[0]], indent=1))
