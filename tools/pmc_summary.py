#!/usr/bin/env python3
"""Summarises the passes of tools/pmc_collect.sh: per kernel name, the average of every counter per dispatch and the average
duration; derived: HBM traffic (FETCH_SIZE x2 per the gfx950 correction of MI355X_MICROARCH.md, + WRITE_SIZE), MFMA busy
fraction.  Usage: pmc_summary.py OUTDIR"""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> values (per dispatch)
dur = defaultdict(list)
for f in sorted(glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    per = defaultdict(float)
    meta = {}
    for r in csv.DictReader(open(f)):
        kname = r["Kernel_Name"] + " grid=" + r.get("Grid_Size", "?")
        key = (r["Dispatch_Id"], kname, r["Counter_Name"])
        per[key] += float(r["Counter_Value"])
        meta[r["Dispatch_Id"]] = (kname, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size", ""))
    for (d, k, cn), v in per.items():
        acc[k][cn].append(v)
    for d, (k, t, g) in meta.items():
        dur[k].append(t)


def short(nm):
    nm = nm.replace("eig::(anonymous namespace)::", "").replace("eig::", "").replace("void ", "")
    for a, b in (("gemm_fast_kernel", "gemmF"), ("cplx", "z"), ("double", "d"), (", ", ",")):
        nm = nm.replace(a, b)
    g = nm.split(" grid=")[-1] if " grid=" in nm else ""
    return (nm.split("(")[0][:56] + " g" + g)[:70]


print("# rocprofv3 --kernel-trace --pmc passes over tools/pmc_targets.py (per-dispatch averages; durations are inflated by counter collection)")
for k in sorted(acc, key=lambda k: -sum(dur[k])):
    if not any(s in k for s in ("eig::",)):
        continue
    c = {cn: sum(v) / len(v) for cn, v in acc[k].items()}
    n = max(len(v) for v in acc[k].values())
    line = "%-72s dispatches/pass %4d  avg duration %9.1f us" % (short(k), n, sum(dur[k]) / len(dur[k]) / 1e3)
    print(line)
    for cn in sorted(c):
        print("      %-34s %16.1f" % (cn, c[cn]))
    if "FETCH_SIZE" in c:
        fb = c["FETCH_SIZE"] * 1024 * 2
        wb = c.get("WRITE_SIZE", 0.0) * 1024
        print("      -> HBM read bytes (FETCH_SIZE KB x 1024 x 2)  %14.0f   write bytes %14.0f" % (fb, wb))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("SQ_BUSY_CYCLES", 0) > 0:
        # SQ_BUSY_CYCLES is summed over the 8 XCCs x their shader engines, MFMA busy over the SIMDs: report the raw ratio and
        # the MOPS count (512 flops per fp64 MOP: calibrated on zgemm 4096^3 = 5.5e11 flops)
        print("      -> MFMA busy cycles / SQ busy cycles = %.3f   fp64 MFMA flops (MOPS x 512) = %.4e" % (
            c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"], c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) * 512))
    if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES", 0) > 0:
        print("      -> wave cycles: waiting %.2f  issue-stalled %.2f  issuing %.2f" % (
            c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"],
            c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]))
