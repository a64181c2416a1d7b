for rep in 1 2; do for v in 250 300; do
CPD_TUNE=1 CPD_GC_WINDOW_MIN64=$v python bench.py --frames 1 --streams 1 --steps 300 --warmup 30 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('min64w=$v', d['ms_per_step'])"
done; done
