#!/usr/bin/env python3
"""Summarises the counter passes of tools/pmc_collect.sh as ONE TABLE: per kernel (name + grid), launches, average duration,
MFMA pipe occupancy, wave-state split and HBM traffic (next to the algorithmic bytes where the launch is a known target).

  MFMA pipe %   = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (SQ_BUSY_CYCLES / 32 shader engines): the share of the launch's
                  cycles in which a SIMD's matrix pipe was busy (zgemm 4096^3 calibrates it: 16 busy cycles per MOP).
  TF/s          = SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 flop / duration (durations of counter passes run ~5-10 % long).
  iss/stall/wait= SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY, SQ_WAIT_ANY over SQ_WAVE_CYCLES.
  HBM MB        = FETCH_SIZE x 2 (gfx950 correction for 16-B/lane streams, MI355X_MICROARCH.md) + WRITE_SIZE, per launch.
Usage: pmc_summary.py OUTDIR [hemv_traffic.json]"""
import csv
import glob
import json
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
        meta[r["Dispatch_Id"]] = (kname, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for (d, k, cn), v in per.items():
        acc[k][cn].append(v)
    for d, (k, t) in meta.items():
        dur[k].append(t)


def short(nm):
    nm = nm.replace("eig::(anonymous namespace)::", "").replace("eig::", "").replace("void ", "")
    for a, b in (("gemm_fast_kernel", "gemmF"), ("cplx", "z"), ("double", "d"), (", ", ",")):
        nm = nm.replace(a, b)
    g = nm.split(" grid=")[-1] if " grid=" in nm else ""
    return nm.split("(")[0][:44], g


# algorithmic bytes of the launches that are stand-alone targets of tools/pmc_targets.py (name fragment, grid size) -> (bytes, label)
N = 4096
KNOWN = {
    # (round 4: the mat-vec grid is one workgroup per CU, the products launch 1-D grids of 64-tile super-tiles -- blas3.hip tile_of)
    # (round 6: the mat-vec kernel has a data-path template argument; the complex 64 x 64 tiles run on gemm_dma_kernel)
    ("panel_mv_kernel<z,1,", str(256 * 320)): (16 * N * (N + 1) // 2, "hemv n=4096: s n(n+1)/2"),
    ("panel_mv_kernel<d,1,", str(256 * 320)): (8 * 2048 * 2049 // 2, "symv n=2048: s n(n+1)/2"),
    ("gemm_dma_kernel<false,16,2>", str(64 * 64 * 256)): (3 * 16 * N * N, "zgemm 4096^3: A + B + C once"),
    ("gemm_dma_kernel<false,8,3>", str(2176 * 256)): (2 * 16 * N * 64 + 2 * 16 * N * (N + 1) // 2, "zher2k k=64: V, W + upper(C) read and written"),
}

rows = []
small = defaultdict(lambda: {"c": defaultdict(list), "dur": [], "n": 0})     # per name: the (name, grid) entries below the cut
CUT = 150.0
for k in acc:
    if "eig::" not in k:
        continue
    nm, g = short(k)
    ncall = max(len(v) for v in acc[k].values())
    us = sum(dur[k]) / len(dur[k]) / 1e3
    if us * ncall < CUT and "panel_mv" not in nm:
        sm_ = small[nm]
        for cn, v in acc[k].items():
            sm_["c"][cn].extend(v)
        sm_["dur"].extend(dur[k])
        sm_["n"] += ncall
        continue
    c = {cn: sum(v) / len(v) for cn, v in acc[k].items()}
    rows.append((us * ncall, nm, g, ncall, us, c))
for nm, sm_ in small.items():       # launches of one kernel over many grids (the Cholesky block rows, shrinking updates): one row
    us = sum(sm_["dur"]) / len(sm_["dur"]) / 1e3
    if us * sm_["n"] < CUT:
        continue
    c = {cn: sum(v) / len(v) for cn, v in sm_["c"].items()}
    rows.append((us * sm_["n"], nm, "(various)", sm_["n"], us, c))
rows.sort(key=lambda r: -r[0])
hdr = "%-46s %9s %6s %10s %7s %7s   %-17s %10s %10s" % ("kernel", "grid", "calls", "avg us", "MFMA %", "TF/s", "iss/stall/wait", "HBM MB", "algo MB")
print("# rocprofv3 --kernel-trace --pmc passes over tools/pmc_targets.py: per-dispatch averages (see the header of tools/pmc_summary.py)")
print(hdr)
hemv = None
for tot, nm, g, ncall, us, c in rows:
    mf = tf = None
    if c.get("SQ_BUSY_CYCLES", 0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        mf = (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (c["SQ_BUSY_CYCLES"] / 32.0) * 100.0
        tf = c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) * 512.0 / (us * 1e-6) * 1e-12
    ws = ""
    if c.get("SQ_WAVE_CYCLES", 0) > 0:
        w = c["SQ_WAVE_CYCLES"]
        ws = "%.2f/%.2f/%.2f" % (c.get("SQ_ACTIVE_INST_ANY", 0) / w, c.get("SQ_WAIT_INST_ANY", 0) / w, c.get("SQ_WAIT_ANY", 0) / w)
    hb = None
    if "FETCH_SIZE" in c:
        hb = (c["FETCH_SIZE"] * 1024 * 2 + c.get("WRITE_SIZE", 0.0) * 1024) / 1e6
    algo = ""
    for (frag, gg), (b, label) in KNOWN.items():
        if nm.startswith(frag) and g == gg:
            algo = "%10.1f" % (b / 1e6)
            if "panel_mv_kernel<z" in nm and hb is not None:
                hemv = {"kernel": "panel_mv_kernel<cplx> (plain hemv launch, n=4096, %d launches averaged)" % ncall,
                        "collected": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/pmc_collect.sh over tools/pmc_targets.py)",
                        "FETCH_SIZE_raw_KB": c["FETCH_SIZE"], "WRITE_SIZE_raw_KB": c.get("WRITE_SIZE", 0.0),
                        "correction": "FETCH_SIZE x2 (gfx950 reports half the bytes of 16-B/lane coalesced streams, MI355X_MICROARCH.md); WRITE_SIZE uncorrected",
                        "fetch_bytes": c["FETCH_SIZE"] * 2048, "write_bytes": c.get("WRITE_SIZE", 0.0) * 1024, "algo_bytes": float(b),
                        "traffic_over_algorithmic": (c["FETCH_SIZE"] * 2048 + c.get("WRITE_SIZE", 0.0) * 1024) / b,
                        "read_over_algorithmic": c["FETCH_SIZE"] * 2048 / b}
    print("%-46s %9s %6d %10.1f %7s %7s   %-17s %10s %10s" % (
        nm, g, ncall, us, "%.1f" % mf if mf is not None and mf > 0.05 else "", "%.1f" % tf if tf else "", ws,
        "%.1f" % hb if hb is not None else "", algo))
if hemv and len(sys.argv) > 2:
    json.dump(hemv, open(sys.argv[2], "w"), indent=1)
