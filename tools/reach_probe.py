#!/usr/bin/env python3
"""Back-trace reach (cells per step, per axis) of the smoke flow: sizes the advection halo of the z-slab decomposition.
    python tools/reach_probe.py [X] [Z] [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fluidx12_amd as fx

X = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Z = int(sys.argv[2]) if len(sys.argv) > 2 else X
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
f = fx.Fluid()
assert f.Init(800, 800, (X, X, Z), jacobi_iters=40)
dt = np.float32(2.0 / X)
dims = (X, X, Z)
worst = [0.0, 0.0, 0.0]
for k in range(steps):
    f.UpdateFrame(dt, k % 3)
    f.Simulate(k % 3)
    if k % 20 == 19:
        f.Synchronize()
        u = f.download(fx.FIELD_VELOCITY)
        r = [float(np.abs(u[a]).max() * dt * dims[a]) for a in range(3)]
        worst = [max(a, b) for a, b in zip(worst, r)]
        print(k + 1, [round(v, 2) for v in r], flush=True)
print("grid %dx%dx%d worst reach (cells) x,y,z = %s" % (X, X, Z, [round(v, 2) for v in worst]))
