for g in 1 2 4; do for k in 20 60; do
python bench.py --headline-only --pairs-per-submission $g --steps $k --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('G=$g K=$k', round(d['ms_per_step'],4), d['stage_ms'])"
done; done
