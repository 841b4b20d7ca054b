#!/bin/bash
# usage (build container, after scripts/profiles_r04.sh ran on the GPU box and gpurun merged
# gpurun_out/r04 back): copies / summarises what is tracked under profiles/
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r04
for f in bf16_cfg2 bf16_cfg2_perblock bf16_cfg2_headlaunch f32_cfg2 bf16_cfg3 bf16_cfg4 fp8_cfg4 bf16_cfg5 fp8_cfg5 bf16_cfg2_B1024 bf16_fst bf16_3st; do
  [ -s $O/${f}_bench.json ] && tail -n 1 $O/${f}_bench.json > profiles/r04_${f}_bench.json
done
[ -s $O/infer.txt ] && cp $O/infer.txt profiles/r04_infer.txt
cp $O/stats_cfg2/*/*kernel_stats.csv profiles/r04_bf16_cfg2_kernel_stats.csv
cp $O/stats_cfg4/*/*kernel_stats.csv profiles/r04_bf16_cfg4_kernel_stats.csv
cp $O/stats_fst/*/*kernel_stats.csv profiles/r04_bf16_fst_kernel_stats.csv
cp $O/stats_3st/*/*kernel_stats.csv profiles/r04_bf16_3st_kernel_stats.csv
[ -d $O/sqfst ] && (echo "# SQ counters of the FST step (bench.py --config fst --no-graph, 3 steps), one rocprofv3 --pmc pass per counter group; kernel time from a --kernel-trace --stats pass of the same (un-captured) workload"; python3 scripts/sq_summary.py $O/sqfst) > profiles/r04_sq_fst.txt
python3 scripts/pmc_summary.py profiles/r04_hbm_traffic.json > /dev/null
(echo "# SQ counters of the cfg4 step (bench.py --config cfg4 --no-graph, 3 steps), one rocprofv3 --pmc pass per counter group; kernel time from the --kernel-trace --stats pass of the same (un-captured) workload"
 python3 scripts/sq_summary.py $O/sqcfg4) > profiles/r04_sq_cfg4.txt
(echo "# SQ counters of the cfg2 step (bench.py --no-graph, 6 steps), one rocprofv3 --pmc pass per counter group; kernel time from a --kernel-trace --stats pass of the same (un-captured) workload"
 python3 scripts/sq_summary.py $O/sqcfg2) > profiles/r04_sq_cfg2.txt
for v in ab; do
  (echo "# d=256 many-queries block forward, B=128 sets, bf16 in/out, inference ($v: ab = k_isab1_fwd256_ab, the producer / consumer wave groups of round 3; onerole = k_isab1_fwd256 of round 2, PCA_D256_AB=0): whole-call times (HIP events), rocprofv3 kernel-trace averages per N, SQ counters (N=2048)"
   grep "whole call" $O/fwd256_$v/bench.log
   python3 - $O/fwd256_$v <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/trace/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
ks = [r for r in rows if re.search(r'isab1_fwd256', r['Kernel_Name'])]
def short(name):
    m = re.search(r"(k_[a-z0-9_]+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]
d = [(int(k['End_Timestamp']) - int(k['Start_Timestamp'])) / 1e3 for k in ks]
h = len(d) // 2
flops = {2048: 2.0 * 128 * 2048 * (2 * 256 * 256 + 2 * 32 * 256), 4096: 2.0 * 128 * 4096 * (2 * 256 * 256 + 2 * 32 * 256)}
for N, part in ((2048, d[:h]), (4096, d[h:])):
    a = sum(part) / len(part)
    print(f"kernel-trace: {short(ks[0]['Kernel_Name'])}: N={N} avg {a:.1f} us ({len(part)} launches) = "
          f"{flops[N] / a / 1e6:.0f} TFLOP/s = {100 * flops[N] / a / 1e6 / 2500:.1f} % of the 2.5 PFLOP/s bf16 MFMA peak")
PY
   python3 scripts/sq_summary.py $O/fwd256_$v k_isab1) > profiles/r04_fwd256_${v}_sq.txt
done
(echo "# d -> d ISAB forward (mab0 + mab1) at B=128, N=2048, d=256, m=32, bf16 activations: scripts/isab256_fwd_bench.py under rocprofv3 --kernel-trace --stats"
 cat $O/isab256/bench.log | grep "ISAB"
 python3 scripts/trace_by_grid.py $O/isab256/trace 13 12) > profiles/r04_isab256_fwd.txt
(echo "# d -> d ISAB as a TRAINING unit (forward with saves + backward incl. every weight gradient) at B=128, N=2048, d=256, m=32, bf16 activations: scripts/isab256_fwdbwd_bench.py under rocprofv3 --kernel-trace --stats; 3.662 GFLOP per set forward + backward (SURVEY.md 8d)"
 cat $O/isab256fb/bench.log | grep "ISAB"
 python3 scripts/trace_by_grid.py $O/isab256fb/trace 39 30) > profiles/r04_isab256_fwdbwd.txt
echo "copied"
