# command measured by ab_libs.sh (edit to taste): the configs[1] bench line, mean and median of the windows
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg2', d['ms_per_step'], d['windows']['ms_per_step_median'])"
