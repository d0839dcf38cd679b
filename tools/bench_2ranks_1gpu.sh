#!/bin/bash
# Dev tool: the --gpus 2 logic of bench.py with two ranks sharing GPU 0 (gloo collectives) - validates the multi-rank
# code path on a one-GPU box; the numbers mean nothing (the ranks compete for the same GPU).
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 WORLD_SIZE=2 LOCAL_RANK=0 KARIOS_BENCH_BACKEND=gloo
RANK=1 python3 bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline > /tmp/bench_rank1.log 2>&1 &
P=$!
RANK=0 python3 bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline
wait $P; echo "rank1 exit $?"; tail -2 /tmp/bench_rank1.log
