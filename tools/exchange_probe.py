#!/usr/bin/env python3
"""What does the device-side block exchange (karios_amd.parallel.RankBlockExchange) cost per step of the streamed loop?  The loop of
bench.py's N > 1 headline on a one-rank RCCL group, with the exchange's stages switched on one by one.

    python tools/exchange_probe.py [steps]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from karios_amd import synth  # noqa: E402
from karios_amd._lib import Context  # noqa: E402
from karios_amd.core import KLTConfiguration  # noqa: E402
from karios_amd.parallel import RankBlockExchange  # noqa: E402
from karios_amd.resident import ResidentPair  # noqa: E402
from karios_amd.stream import FrameStream  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 200
as_json = "--json" in sys.argv
S = 10980
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
mon_t, ref_t = synth.make_pair_torch(S, S, 0.5, 0.25, device=dev)
torch.cuda.synchronize()
ctx = Context(0)
pair = ResidentPair.from_device_pointers(mon_t.data_ptr(), ref_t.data_ptr(), np.uint16, S, S, ctx=ctx, keepalive=(mon_t, ref_t))
conf = KLTConfiguration()


LAG = "--at-submit" not in sys.argv        # issue the exchange when the step is COLLECTED (bench.py's way) instead of at submission
DEPTH = 2
BATCH = next((int(a.split("=", 1)[1]) for a in sys.argv if a.startswith("--batch=")), 4)
if BATCH < DEPTH:
    sys.exit(f"--batch={BATCH}: a send slot is only free once its batch has been gathered, which happens when the batch's last step is "
             f"COLLECTED ({DEPTH} submissions later): batch >= {DEPTH} (RankBlockExchange.arm refuses anything else)")


def run(parts):
    ex = RankBlockExchange(ctx, conf.maxCorners, True, device=dev, parts=parts, batch=BATCH) if parts is not None else None
    pend_of = {}
    with FrameStream(0.4, depth=DEPTH) as stream:
        def collected(results):
            if ex is not None and LAG:
                for d in results:
                    ex.issue(d.tag, pend_of.pop(d.tag))

        def one(k):
            if ex is None:
                return stream.submit(pair, conf)
            ex.arm(k)
            if LAG:
                collected(stream.submit(pair, conf, tag=k, on_submitted=lambda p, k=k: pend_of.__setitem__(k, p)))
            else:
                stream.submit(pair, conf, on_submitted=lambda p, k=k: ex.issue(k, p))
        for k in range(30):
            one(k)
        collected(stream.drain()); ctx.sync(); torch.cuda.synchronize()
        marks = [time.perf_counter()]
        for k in range(steps):
            one(30 + k)
            marks.append(time.perf_counter())
        collected(stream.drain())
        rows = ex.finish() if ex is not None else None
        ctx.sync(); torch.cuda.synchronize()
        dt = time.perf_counter() - marks[0]
    gaps = sorted(b - a for a, b in zip(marks, marks[1:]))
    if not as_json:
        print(f"{str(parts):32s} {dt / steps * 1e3:.4f} ms per step, median submit interval {gaps[len(gaps) // 2] * 1e3:.4f} ms, rows {rows}", flush=True)
    return dt / steps * 1e3, gaps[len(gaps) // 2] * 1e3, rows


if as_json:
    # A/B inside ONE process on one box (two processes differ by 1 - 2 % on this pool): plain loop and full exchange alternate; the
    # first round of each pays the one-time costs (communicator, first launches) and is dropped
    import json
    FULL = "sink,wait,gather,account"
    res = {"plain": [], "exchange": []}
    for rep in range(4):
        for key, parts in (("plain", None), ("exchange", FULL)):
            ms, med, rows = run(parts)
            if rep:
                res[key].append({"ms_per_step": ms, "median_submit_interval_ms": med, "rows": rows})
    out = {"steps": steps, "plain_ms_per_step": min(r["ms_per_step"] for r in res["plain"]), "exchange_ms_per_step": min(r["ms_per_step"] for r in res["exchange"]),
           "plain_median_ms": min(r["median_submit_interval_ms"] for r in res["plain"]), "exchange_median_ms": min(r["median_submit_interval_ms"] for r in res["exchange"]),
           "rows_per_run": [r["rows"] for r in res["exchange"]], "runs": res}
    out["ratio_ms_per_step"] = out["exchange_ms_per_step"] / out["plain_ms_per_step"]
    out["ratio_median"] = out["exchange_median_ms"] / out["plain_median_ms"]
    print(json.dumps(out))
else:
    for rep in range(2):
        for parts in (None, "sink", "sink,wait", "sink,wait,gather", "sink,wait,gather,account", None):
            run(parts)
dist.destroy_process_group()
