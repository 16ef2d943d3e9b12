#!/usr/bin/env python3
"""Condense the rocprofv3 output that tools/profile_round.sh left under gpurun_out/ into the
tracked summaries under profiles/:  <tag>_kernel_stats.csv (the --stats table), <tag>_pmc.csv
(per-kernel counter sums and per-launch averages) and profiles/traffic.json (HBM bytes per launch of
the dominant kernel, corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE is in KiB and, on
gfx950, reports half of the bytes of wide coalesced reads — the per-lane 16-byte node/triangle
gathers here are NOT that pattern, so both the raw and the doubled figure are recorded)."""
import glob
import json
import os
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
config = sys.argv[2] if len(sys.argv) > 2 else "config3"                  # key in profiles/traffic.json
dominant = sys.argv[3] if len(sys.argv) > 3 else "k_bvh<1, false>"        # the kernel bench.py's roofline is about (prefix match)
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)


def short(name):
    import re
    m = re.search(r"(k_\w+<[\w, ]+>|k_\w+)", name)
    return m.group(1) if m else name[:40]


# a tag profiled twice leaves both runs' files in gpurun_out/ (gpurun merges): take the newest
st = sorted(glob.glob(os.path.join(G, f"{tag}_stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
if st:
    df = pd.read_csv(st[-1])
    df["Name"] = df["Name"].map(short)
    df.to_csv(os.path.join(P, f"{tag}_kernel_stats.csv"), index=False)
    print(df[["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"]].to_string(index=False))
    log = os.path.join(G, f"{tag}_stats.log")
    if os.path.exists(log):
        lines = [l for l in open(log) if l.startswith("{")]
        if lines:
            open(os.path.join(P, f"{tag}_bench_under_rocprof.json"), "w").write(lines[-1])

rows = []
for sub in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq", "pmc_sq2", "pmc_ta"):
    f = sorted(glob.glob(os.path.join(G, f"{tag}_{sub}", "*", "*_counter_collection.csv")), key=os.path.getmtime)
    if not f:
        continue
    df = pd.read_csv(f[-1])
    df["k"] = df["Kernel_Name"].map(short)
    g = df.groupby(["k", "Counter_Name"])["Counter_Value"].agg(["sum", "count"]).reset_index()
    rows.append(g)
if rows:
    allc = pd.concat(rows)
    allc["per_launch"] = allc["sum"] / allc["count"]
    allc = allc[allc["k"].str.startswith("k_")]
    allc.to_csv(os.path.join(P, f"{tag}_pmc.csv"), index=False)
    piv = allc.pivot_table(index="k", columns="Counter_Name", values="per_launch")
    print(piv.to_string())
    match = [x for x in piv.index if x == dominant] or [x for x in piv.index if x.startswith(dominant)]
    if not match or "FETCH_SIZE" not in piv.columns:
        sys.exit(f"summarize_profile: dominant kernel {dominant!r} (or FETCH_SIZE) is not in the counter table: traffic.json left as it was")
    k = match[0]
    if True:
        def col(name):
            return float(piv.loc[k, name]) if name in piv.columns and pd.notna(piv.loc[k, name]) else None
        fetch_kib = col("FETCH_SIZE")
        write_kib = col("WRITE_SIZE") or 0.0
        # FETCH_SIZE correction for THIS kernel's access pattern: per-lane gathers of 16-byte pieces of 64-byte (node) and
        # 48-byte (triangle) records.  tools/fetch_calib.hip gathers a known number of such records from a table larger
        # than the Infinity Cache under rocprofv3 --pmc FETCH_SIZE; profiles/fetch_calib.json holds reported / true bytes.
        factor, fnote = 1.0, "FETCH_SIZE used raw (no calibration file)"
        cpath = os.path.join(P, "fetch_calib.json")
        if os.path.exists(cpath):
            cal = json.load(open(cpath))
            rep = cal.get("gather64", {}).get("reported_over_true")
            if rep:
                factor = 1.0 / rep
                fnote = (f"FETCH_SIZE x {factor:.3f}: calibrated on 64-byte random record gathers (4 x 16 B per lane) from a "
                         f"{cal.get('table_mib', '?')} MiB table (profiles/fetch_calib.json)")
        units = None
        bj = os.path.join(P, f"{tag}_bench_under_rocprof.json")
        if os.path.exists(bj):
            try:
                units = float(json.load(open(bj))["roofline"]["rays_per_launch"])
            except Exception:
                units = None
        lanes, insts, grbm = col("SQ_THREAD_CYCLES_VALU"), col("SQ_INSTS_VALU"), col("GRBM_GUI_ACTIVE")
        ta = col("TA_BUSY_avr")
        simd_quads = grbm / 8.0 / 4.0 * 1024.0 if grbm else None   # GRBM_GUI_ACTIVE sums the 8 XCDs; one VALU instruction = one quad-cycle of its SIMD
        valu_busy = insts / simd_quads if insts and simd_quads else None
        ta_busy = ta / (grbm / 8.0) if ta and grbm else None
        hbm = fetch_kib * 1024 * factor + write_kib * 1024
        bound = "valu"
        if ta_busy and valu_busy and ta_busy > valu_busy:
            bound = "ta/l1"
        out = {
            "kernel": k, "tag": tag, "units_per_launch": units,
            "fetch_bytes_per_launch_raw": fetch_kib * 1024, "write_bytes_per_launch": write_kib * 1024, "fetch_factor": round(factor, 4),
            "bytes_per_launch": hbm,
            "hbm_bytes_per_unit": hbm / units if units else None,
            "valu_lane_insts_per_launch": lanes, "valu_insts_per_launch": insts,
            "valu_lane_insts_per_unit": lanes / units if lanes and units else None,
            "valu_busy": round(valu_busy, 4) if valu_busy else None,
            "valu_active_lanes": round(lanes / (64.0 * insts), 4) if lanes and insts else None,
            "ta_busy": round(ta_busy, 4) if ta_busy else None,
            "bound": bound, "bound_busy": round(ta_busy if bound == "ta/l1" else (valu_busy or 0.0), 4),
            "note": "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ | TA, one run each), KiB -> bytes, averaged over the "
                    "launches of one frame (steps 1, warmup 1); " + fnote,
        }
        path = os.path.join(P, "traffic.json")
        allt = {}
        if os.path.exists(path):
            try:
                allt = json.load(open(path))
            except Exception:
                allt = {}
        if "kernel" in allt:   # round-1 layout: one record
            allt = {"config3_r01": allt}
        allt[config] = out
        json.dump(allt, open(path, "w"), indent=1)
        print(json.dumps(out, indent=1))
