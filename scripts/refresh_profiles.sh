#!/bin/bash
# usage (on the GPU box, through gpurun): scripts/refresh_profiles.sh
# Regenerates everything under profiles/ that the bench line cites: the default bench run, the
# rocprofv3 kernel statistics of the same command, the PMC traffic passes, the fp32 run.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_bf16.json 2> $O/bench_bf16.err
tail -1 $O/bench_bf16.json | cut -c1-400
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline > $O/bench_bf16_prof.json 2> $O/bench_bf16_prof.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --mode bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $R/gpurun_out/pmc_$c.log 2>&1
done
python3 $R/bench.py --mode f32 --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err
python3 $R/bench.py --config cfg3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
echo refreshed
