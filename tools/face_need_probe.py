#!/usr/bin/env python3
"""How many advect-halo planes each slab face actually needs, step by step (tools/reach_probe.py gives the global maximum,
which sizes the allocation; this shows what an exchange that follows the flow would send).
    python tools/face_need_probe.py N [steps]        # N = bench.py's weak-scaling rank count (grid from bench.workload_grid)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
import fluidx12_amd as fx

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 140
(X, Y, Z), halo = bench.workload_grid(256, N, "weak")
f = fx.Fluid()
assert f.Init(800, 800, (X, Y, Z), jacobi_iters=40)
dt = np.float32(2.0 / Y)
faces = [r * Z // N for r in range(1, N)]
for k in range(steps):
    f.UpdateFrame(dt, k % 3)
    f.Simulate(k % 3)
    if k % 20 == 19 or k == steps - 1:
        f.Synchronize()
        uz = f.download(fx.FIELD_VELOCITY)[2]
        up = (np.maximum(-uz, 0).max(axis=(1, 2)) * dt * Z)       # cells a voxel of plane z reaches towards +z (pos - u dt)
        dn = (np.maximum(uz, 0).max(axis=(1, 2)) * dt * Z)
        need = []
        for zf in faces:
            d = np.arange(0, 40)
            lo = max((up[zf - 1 - i] - i) for i in d if zf - 1 - i >= 0)          # the lower rank reading planes >= zf
            hi = max((dn[zf + i] - i) for i in d if zf + i < Z)                   # the upper rank reading planes < zf
            need.append(int(np.ceil(max(lo, hi, 0))) + 1)                         # + the second trilinear tap
        print("step %4d  planes needed per face %s   (allocated %d; global reach %.1f)" % (k + 1, need, halo, max(up.max(), dn.max())), flush=True)
