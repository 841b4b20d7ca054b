#!/bin/bash
# usage (GPU box): scripts/ab_env.sh <config> <steps> "ENV1=a ENV2=b" "ENV1=c" ...   -- same-box A/B of bench.py
CFG=$1; STEPS=$2; shift 2
for E in "$@"; do
  R=$(env $E python bench.py --config $CFG --steps $STEPS --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['windows']['ms_per_step_median'], d['value'])")
  echo "[$E] $R"
done
