for c in 1 2 3; do for k in 20 60; do
python bench.py --headline-only --contexts $c --steps $k --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('contexts=$c K=$k', round(d['ms_per_step'],4), d['roofline'].get('kernel_ms'), d['roofline'].get('kernel_ms_alone'))"
done; done
