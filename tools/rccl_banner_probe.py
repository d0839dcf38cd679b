#!/usr/bin/env python3
"""Does this RCCL build print its version banner on stdout, and which environment setting silences it?  (bench.py must end with ONE
JSON line.)  python tools/rccl_banner_probe.py"""
import os, subprocess, sys
code = '''
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29581")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
dist.destroy_process_group(); print("done")
'''
for env in ({}, {"RCCL_LOG_LEVEL": "0"}, {"NCCL_DEBUG": "NONE"}, {"RCCL_LOG_LEVEL": "0", "NCCL_DEBUG": "NONE"}, {"NCCL_DEBUG": "WARN"}):
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    print(env, "-> stdout lines:", [l for l in r.stdout.splitlines()][:8], "| banner on stderr:", "Librccl" in r.stderr)
