"""`python bench.py --gpus N` without a launcher environment: spawn the N ranks (before anything touches the GPU)."""
from __future__ import annotations

import os
import subprocess
import sys


def launch_ranks(a) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    in their environment), relay rank 0's JSON line, fail when any rank fails.  This parent never imports torch or touches
    HIP: a process that has initialised the GPU must not be replaced or forked on this pool."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KARIOS_BENCH_LAUNCHER="bench.py (spawned ranks)")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(sys.argv[0])] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    out0 = ""
    failed = None
    try:
        out0, _ = procs[0].communicate()
        for r, p in enumerate(procs):
            rc = p.wait()
            if rc != 0 and failed is None:
                failed = (r, rc)
    except BaseException:
        failed = failed or (-1, 1)
        raise
    finally:
        if failed is not None:
            for p in procs:                      # exactly the processes started above
                if p.poll() is None:
                    p.terminate()
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if failed is not None:
        sys.stderr.write(out0)
        print(f"bench.py: rank {failed[0]} exited with status {failed[1]}", file=sys.stderr)
        return failed[1] or 1
    if not lines:
        sys.stderr.write(out0)
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
        return 1
    # rank 0's detail lines first (stdout, in order), its JSON line LAST
    for ln in out0.splitlines():
        if ln is not lines[-1] and ln != lines[-1]:
            print(ln)
    print(lines[-1], flush=True)
    return 0
