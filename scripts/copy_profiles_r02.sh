#!/bin/bash
# usage (build container, after scripts/profiles_r02.sh ran on the GPU box and gpurun merged
# gpurun_out/r02 back): copies / summarises what is tracked under profiles/
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r02
for f in bf16_cfg2 f32_cfg2 bf16_cfg3 bf16_cfg4 fp8_cfg4 bf16_cfg5 fp8_cfg5 bf16_cfg2_B1024; do
  tail -n 1 $O/${f}_bench.json > profiles/r02_${f}_bench.json
done
cp $O/stats_cfg2/*/*kernel_stats.csv profiles/r02_bf16_cfg2_kernel_stats.csv
cp $O/stats_cfg4/*/*kernel_stats.csv profiles/r02_bf16_cfg4_kernel_stats.csv
python3 scripts/pmc_summary.py profiles/r02_hbm_traffic.json > /dev/null
(echo "# SQ counters of the cfg4 step (bench.py --config cfg4 --no-graph, 3 steps), one rocprofv3 --pmc pass per counter group; kernel time from the --kernel-trace --stats pass of the same config (profiles/r02_bf16_cfg4_kernel_stats.csv)"
 python3 scripts/sq_summary.py $O/sqcfg4) > profiles/r02_sq_cfg4.txt
(echo "# SQ counters of the cfg2 step (bench.py --no-graph, 6 steps), one rocprofv3 --pmc pass per counter group; kernel time from a --kernel-trace --stats pass of the same (un-captured) workload"
 python3 scripts/sq_summary.py $O/sqcfg2) > profiles/r02_sq_cfg2.txt
for v in fused pair; do
  (echo "# d=256 many-queries block forward, B=128 sets, bf16 in/out, inference ($v): whole-call times (HIP events), rocprofv3 kernel-trace averages per N, SQ counters (N=2048)"
   grep "whole call" $O/fwd256_$v/bench.log
   python3 - $O/fwd256_$v <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/trace/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
ks = [r for r in rows if re.search(r'isab1_fwd256|k_mab1_fwd|k_rowgemm', r['Kernel_Name'])]
def short(name):
    m = re.search(r"(k_[a-z0-9_]+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]
names = []
for k in ks:
    n = short(k['Kernel_Name'])
    if n not in names:
        names.append(n)
flops = {2048: 2.0 * 128 * 2048 * (2 * 256 * 256 + 2 * 32 * 256), 4096: 2.0 * 128 * 4096 * (2 * 256 * 256 + 2 * 32 * 256)}
tot = {2048: 0.0, 4096: 0.0}
for n in names:
    d = [(int(k['End_Timestamp']) - int(k['Start_Timestamp'])) / 1e3 for k in ks
         if short(k['Kernel_Name']) == n]
    h = len(d) // 2
    a, b = sum(d[:h]) / h, sum(d[h:]) / (len(d) - h)
    tot[2048] += a
    tot[4096] += b
    print(f"kernel-trace: {n}: N=2048 avg {a:.1f} us ({h} launches), N=4096 avg {b:.1f} us")
for N in (2048, 4096):
    print(f"kernel time N={N}: {tot[N]:.1f} us = {flops[N] / tot[N] / 1e6:.0f} TFLOP/s = "
          f"{100 * flops[N] / tot[N] / 1e6 / 2500:.1f} % of the 2.5 PFLOP/s bf16 MFMA peak")
PY
   python3 scripts/sq_summary.py $O/fwd256_$v) > profiles/r02_fwd256_${v}_sq.txt
done
echo "copied"
