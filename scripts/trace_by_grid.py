#!/usr/bin/env python3
"""Aggregate a rocprofv3 kernel trace by (kernel, grid): usage trace_by_grid.py <dir> [steps]"""
import collections, csv, glob, re, sys
d = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_\w+)(<[^>]*>)?", r["Kernel_Name"])
    short = m.group(0) if m else r["Kernel_Name"][:40]
    key = (short, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    agg[key][0] += 1
    agg[key][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f"{k[0]:34s} grid {k[1]:>10s} {k[2]:>5s} {k[3]:>5s} calls/step {v[0]/steps:5.1f} "
          f"avg {v[1]/v[0]/1000:8.1f} us  per-step {v[1]/steps/1000:8.1f} us {100*v[1]/tot:5.1f}%")
