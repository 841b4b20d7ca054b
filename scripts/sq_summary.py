#!/usr/bin/env python3
"""Per-kernel SQ counter table from the passes of scripts/pmc_fwd256.sh / pmc_sq.sh.
usage: python scripts/sq_summary.py <dir with sq_*/ and trace/> [kernel-name substring]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


tab = defaultdict(dict)
for f in sorted(glob.glob(os.path.join(root, "sq_*", "*", "*counter_collection.csv"))):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = (short(r["Kernel_Name"]), r["Counter_Name"])
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
    for (kern, c), v in tot.items():
        tab[kern][c] = v / cnt[(kern, c)]
dur = {}
for f in glob.glob(os.path.join(root, "trace", "*", "*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        dur[short(r["Name"])] = float(r["AverageNs"]) / 1e3
for kern, c in sorted(tab.items(), key=lambda kv: -dur.get(kv[0], 0)):
    if pat not in kern:
        continue
    print(f"== {kern}  avg {dur.get(kern, float('nan')):.1f} us")
    for k in sorted(c):
        print(f"   {k:28s} {c[k]:16.0f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
        # SQ_BUSY_CYCLES is summed over the 32 shader engines (8 XCDs x 4), SQ_VALU_MFMA_BUSY_CYCLES
        # over the 1024 SIMDs: 32 SIMDs per engine
        print(f"   -> MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES) = "
              f"{c['SQ_VALU_MFMA_BUSY_CYCLES'] / max(32 * c['SQ_BUSY_CYCLES'], 1):.3f}")
        # SQ_BUSY_CYCLES: per SE summed; MFMA busy counts per SIMD cycles: report the ratio to wave cycles
        wc = c.get("SQ_WAVE_CYCLES", 0) * 4          # quad-cycles -> cycles
        print(f"   -> MFMA-busy / (4 x SQ_WAVE_CYCLES) = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / max(wc, 1):.3f}"
              f"; WAIT_ANY / WAVE_CYCLES = {c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.3f}"
              f"; WAIT_INST_ANY / WAVE_CYCLES = {c.get('SQ_WAIT_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.3f}")
