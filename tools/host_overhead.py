#!/usr/bin/env python3
"""Host-side cost per iteration of the partitioned BPCG loop (Python + ctypes + kernel launches +
RCCL calls), measured on ONE GPU with a tiny system so that device time is negligible:
rank 0 of a 1-rank group drives DistributedBpcg2.iterate with (a) communication skipped and
(b) the RCCL ctypes communicator forced to issue its all-reduces (1-rank communicator)."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "navier-stokes-solver_amd")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist

import hipla
from distributed import DistributedBpcg2, TorchComm
from rccl_comm import RcclComm
from staggered_grid import mac_stokes

dist.init_process_group("gloo", init_method="file://" + tempfile.mkdtemp() + "/rdv", rank=0, world_size=1)
eng = hipla.get_engine()
s = mac_stokes(3, 12, 0.01)
f, g = s.rhs(0)
for name, comm in (("no-comm", TorchComm(dist, eng)), ("rccl-allreduce", RcclComm(dist, eng))):
    run = DistributedBpcg2(s, f, g, s.line_blocks(3), dist, eng, comm=comm)
    if name != "no-comm":
        comm.size = 2                      # issue the all-reduces (halo plans are empty with one rank)
    n = 2000
    run.start(tol=0.0, maxsteps=n + 10)
    run.iterate(0, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run.iterate(10, n + 10)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("%-16s host issue %.1f us/iteration, wall %.1f us/iteration" % (name, 1e6 * t_issue / n, 1e6 * t_all / n))
    comm.size = 1
ses_loop = run.loop
t0 = time.perf_counter()
ses_loop.enqueue(n + 10, n + 10)   # no-op
n2 = 2000
run.start(tol=0.0, maxsteps=n2)
t0 = time.perf_counter()
ses_loop.enqueue(0, n2)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
print("%-16s host issue %.1f us/iteration, wall %.1f us/iteration" % ("C loop (1 GPU)", 1e6 * t_issue / n2,
                                                                     1e6 * (time.perf_counter() - t0) / n2))
