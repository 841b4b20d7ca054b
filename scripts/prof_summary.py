import csv, glob, re, sys
tag = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 106
import os
f = max(glob.glob(f'gpurun_out/prof_{tag}/*/*kernel_stats.csv'), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    name = r['Name']
    m = re.search(r'(k_[a-z0-9_]+|FillFunctor<\w+>|copyBuffer|Cat\w+)', name)
    nm = m.group(1) if m else name[:30]
    t = re.search(nm + r'(<[^>(]*>)', name)
    calls = int(r['Calls']); avg = float(r['AverageNs']) / 1e3; total = float(r['TotalDurationNs']) / 1e3
    print(f"{nm:20s}{(t.group(1) if t else ''):18s} calls/step {calls/steps:6.2f}  avg {avg:8.1f} us  per-step {total/steps:8.1f} us")
print("sum per step (us):", round(sum(float(r['TotalDurationNs']) for r in rows) / 1e3 / steps, 1),
      "launches/step", round(sum(int(r['Calls']) for r in rows) / steps, 1))
