#!/usr/bin/env python3
"""FLUIDX_COMM_PRIORITY probe: wall time per step of a 4-rank loop-back group (256^3 per rank, overlap 2, rounds of 9) with the
library's HIP-event timing marks on and off.  Run once per priority setting (the variable is read when the group is created)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fluidx12_amd as fx
from fluidx12_amd import capi

G, N = 256, 4
fl = []
for r in range(N):
    f = fx.Fluid()
    assert f.Init(64, 64, (G, G, G * N), slab=(r * G, G), jacobi_iters=40, halo_advect=22, halo_jacobi=9)
    fl.append(f)
fx.comm_init_local(fl)
for f in fl:
    f.set_option(capi.OPT_OVERLAP, int(os.environ.get("PROBE_OVERLAP", "2")))
    f.set_option(capi.OPT_JACOBI_ROUND, 9)
dt = np.float32(2.0 / G)
k = 0
for timing in (False, True, False):
    for f in fl:
        f.timing_enable(timing)
    for _ in range(3):
        fl[0].UpdateFrame(dt, k % 3); fl[0].Simulate(k % 3); k += 1
    fl[0].Synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fl[0].UpdateFrame(dt, k % 3); fl[0].Simulate(k % 3); k += 1
    t_enq = time.perf_counter() - t0                       # the host's share: enqueueing ten steps
    fl[0].Synchronize()
    print("FLUIDX_COMM_PRIORITY=%s overlap %s timing marks %-3s: %.3f ms per step (host enqueue %.3f ms per step)" % (os.environ.get("FLUIDX_COMM_PRIORITY", "0"), os.environ.get("PROBE_OVERLAP", "2"),
          "on" if timing else "off", (time.perf_counter() - t0) / 10 * 1e3, t_enq / 10 * 1e3), flush=True)
    for f in fl:
        f.timing_read(True)
