#!/usr/bin/env python3
"""Step time over the run: blocks of `block` steps from the zero state, each timed between two synchronisations (bench.py's default
workload: 256^3, 40 sweeps, fp32).  Shows what a short timed region (`--steps 20 --warmup 5`) sees against a long one.

    python tools/step_time_profile.py [blocks] [block] [grid]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluidx12_amd as fx

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 24
block = int(sys.argv[2]) if len(sys.argv) > 2 else 5
grid = int(sys.argv[3]) if len(sys.argv) > 3 else 256
repeat = int(sys.argv[4]) if len(sys.argv) > 4 else 1      # > 1: a fresh context from the zero state again, on a GPU that is warm by then
for rep in range(repeat):
  f = fx.Fluid()
  assert f.Init(1920, 1080, (grid, grid, grid), jacobi_iters=40)
  dt = np.float32(f.default_time_step())
  k = 0
  print("-- run", rep)
  for b in range(blocks):
    f.Synchronize()
    t0 = time.perf_counter()
    for _ in range(block):
        f.UpdateFrame(dt, k % 3)
        f.Simulate(k % 3)
        k += 1
    f.Synchronize()
    print("steps %3d..%3d  %.4f ms per step" % (k - block, k - 1, (time.perf_counter() - t0) / block * 1e3), flush=True)
  f.Release()
