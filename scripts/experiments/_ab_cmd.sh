for v in 1024 1536 1280 768; do
PCA_WGRAD_RPW=$v python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg2 rpw=$v', d['ms_per_step'], d['windows']['ms_per_step_median'])"
done
