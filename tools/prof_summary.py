#!/usr/bin/env python3
"""Summarise rocprofv3 output of bench.py into profiles/ (tracked) and profiles/traffic.json.

    python3 tools/prof_summary.py gpurun_out/r01 profiles/r01 headline transform

Expects <src>/stats (rocprofv3 --kernel-trace --stats), <src>/fetch (--pmc FETCH_SIZE) and
<src>/write (--pmc WRITE_SIZE) as produced by the commands recorded in the summary.
Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE
and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide
(16 B/lane) coalesced streaming read, so the read side is doubled for kernels whose loads
are 16-byte streaming loads (k_transform, k_sweep); WRITE_SIZE is exact for 16 B/lane
streaming stores and uncalibrated for the 8 B/lane stores k_transform issues.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    n = name.replace("void ", "").replace("efa::(anonymous namespace)::", "")
    return n.split("(")[0]


def main():
    src, dst, workload, path = sys.argv[1:5]
    cmd = sys.argv[5] if len(sys.argv) > 5 else "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    lines = []
    stats = max(glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")), key=os.path.getmtime)
    rows = list(csv.DictReader(open(stats)))
    lines.append("# rocprofv3 --kernel-trace --stats --output-format csv -- " + cmd)
    lines.append("# (durations in microseconds)")
    lines.append("%-34s %6s %12s %12s %7s" % ("kernel", "calls", "avg_us", "total_us", "pct"))
    avg = {}
    for r in rows:
        k = short(r["Name"])
        avg[k] = float(r["AverageNs"]) / 1e3
        lines.append("%-34s %6s %12.1f %12.1f %7s" % (k, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                  float(r["TotalDurationNs"]) / 1e3, r["Percentage"]))
    pmc = {}
    for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        f = sorted(glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv")), key=os.path.getmtime, reverse=True)
        if not f:
            continue
        acc = defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == ctr:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        pmc[ctr] = dict((k, sum(v) / len(v)) for k, v in acc.items())
    lines.append("")
    lines.append("# PMC passes (separate runs): rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace")
    lines.append("%-34s %16s %16s %18s" % ("kernel", "FETCH_SIZE KiB", "WRITE_SIZE KiB", "HBM bytes/launch*"))
    traffic = {}
    for k in sorted(set(pmc.get("FETCH_SIZE", {})) | set(pmc.get("WRITE_SIZE", {}))):
        fkb = pmc.get("FETCH_SIZE", {}).get(k, float("nan"))
        wkb = pmc.get("WRITE_SIZE", {}).get(k, float("nan"))
        stream = k.startswith("k_transform") or k.startswith("k_sweep") or k.startswith("k_form_perts") or k.startswith("k_gcm")
        b = (2.0 if stream else 1.0) * fkb * 1024 + wkb * 1024
        traffic[k] = b
        lines.append("%-34s %16.1f %16.1f %18.4g" % (k, fkb, wkb, b))
    lines.append("# * read side doubled (gfx950 FETCH_SIZE correction) for the 16 B/lane streaming kernels")
    # further counter passes (pmc1, pmc2, ...): per-kernel averages
    extra = defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(src, "pmc*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
            acc = defaultdict(list)
            for r in csv.DictReader(open(f)):
                acc[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
            for (k, cname), v in acc.items():
                extra[k][cname] = sum(v) / len(v)
    if extra:
        lines.append("")
        lines.append("# further counter passes (one rocprofv3 --pmc run each, averages per launch)")
        for k in sorted(extra):
            if k.startswith("__amd") or k.startswith("k_fill") or k.startswith("k_set"):
                continue
            lines.append(k)
            for cname in sorted(extra[k]):
                lines.append("    %-28s %18.4g" % (cname, extra[k][cname]))
    open(dst + "_summary.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    tj = os.path.join(os.path.dirname(dst) or ".", "traffic.json")
    t = json.load(open(tj)) if os.path.exists(tj) else {}
    kname = [k for k in traffic if k.startswith({"transform": "k_transform", "gc": "k_sweep_gc", "gcm": "k_gcm"}.get(path, "k_sweep<"))]
    if kname:
        t["%s:%s" % (workload, path)] = traffic[kname[0]]
        json.dump(t, open(tj, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
