"""Spawning the ranks of a multi-process test (gloo rendezvous on 127.0.0.1).

`run_world` starts `world` processes running ``worker(rank, world, port, queue, *extra)`` and returns the `world` items the
workers put on the queue.  The port is probed (bind to 0, close) and can be taken by somebody else before rank 0 binds
it, and a worker that dies during initialisation leaves its peers waiting in the rendezvous: both show as a worker exiting
non-zero, and the attempt is repeated with a fresh port (the survivors are terminated by their own process handles).  A
genuine failure inside a worker fails every attempt and then the test."""
import queue as _queue
import socket
import time

import pytest
import torch.multiprocessing as mp


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def run_world(worker, world, extra=(), timeout=500, join_timeout=120, attempts=3):
    last = ""
    for _ in range(attempts):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = free_port()
        procs = [ctx.Process(target=worker, args=(r, world, port, q) + tuple(extra)) for r in range(world)]
        for p in procs:
            p.start()
        items, ok = [], True
        deadline = time.time() + timeout
        try:
            while len(items) < world:
                try:
                    items.append(q.get(timeout=1.0))
                except _queue.Empty:
                    codes = [p.exitcode for p in procs]
                    if any(c not in (None, 0) for c in codes):
                        ok, last = False, f"worker exit codes {codes}"
                        break
                    if time.time() > deadline:
                        ok, last = False, f"no result within {timeout} s (exit codes {codes})"
                        break
            if ok:
                for p in procs:
                    p.join(join_timeout)
                codes = [p.exitcode for p in procs]
                if all(c == 0 for c in codes):
                    return items
                ok, last = False, f"worker exit codes {codes}"
        finally:
            for p in procs:
                if p.is_alive():
                    p.terminate()
                    p.join(10)
    pytest.fail(f"{world}-process run failed {attempts} times: {last}")
