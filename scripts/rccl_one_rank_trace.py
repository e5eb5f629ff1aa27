#!/usr/bin/env python
"""Tuning aid (GPU box): what does RCCL's INFO log look like for the bucket's all-reduce?  One rank (two ranks cannot
share a device under RCCL), NCCL_DEBUG=INFO / INIT,TUNING into a file, then the lines bench.py's `exchange.rccl_trace`
would quote."""
import os
import sys
log = "/tmp/hgs_rccl_trace_test.log"
os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,TUNING,COLL", NCCL_DEBUG_FILE=log, MASTER_ADDR="127.0.0.1",
                  MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
import torch
import torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.zeros(59_000_000, device="cuda")
for _ in range(3):
    dist.all_reduce(x)
torch.cuda.synchronize()
dist.destroy_process_group()
lines = open(log, errors="replace").read().splitlines()
print(len(lines), "log lines; those naming an algorithm / protocol / channels:")
seen = set()
for ln in lines:
    low = ln.lower()
    if ("algo" in low and "proto" in low) or "channels" in low or "version" in low:
        key = ln.split("]")[-1].strip()[:200]
        if key not in seen:
            seen.add(key)
            print("  ", key)
print("first 15 lines:")
for ln in lines[:15]:
    print("  ", ln[:200])
