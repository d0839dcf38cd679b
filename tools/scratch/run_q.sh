python tools/scratch/inflight_q.py 2>&1 | grep n_ctx
python tools/scratch/inflight_q.py 2>&1 | grep n_ctx
bash tools/scratch/run_ab.sh - -
