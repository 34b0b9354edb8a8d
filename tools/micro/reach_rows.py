"""Which share of the 64-voxel rows (= waves of k_advect_lds) back-traces inside a +-W window, per frame: sizes the staged border."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fluidx12_amd as fx
X = int(sys.argv[1]) if len(sys.argv) > 1 else 256
f = fx.Fluid(); assert f.Init(0, 0, (X, X, X), jacobi_iters=40)
dt = np.float32(f.default_time_step())
k = 0
for frame in (25, 60, 132, 250, 400):
    while k < frame:
        f.UpdateFrame(dt, k % 3); f.Simulate(k % 3); k += 1
    f.Synchronize()
    u = f.download(fx.FIELD_VELOCITY)                         # [3][Z][Y][X]
    d = [-(u[a] * dt * X) for a in range(3)]                  # displacement in cells
    def inside(a, w): return (d[a] >= -w) & (d[a] < w)
    rows = lambda m: m.reshape(X, X, X // 64, 64).all(axis=3)
    w1 = rows(inside(0, 1) & inside(1, 1) & inside(2, 1))
    w2xy = rows(inside(0, 2) & inside(1, 2) & inside(2, 1))
    w2y = rows(inside(0, 1) & inside(1, 2) & inside(2, 1))
    w2 = rows(inside(0, 2) & inside(1, 2) & inside(2, 2))
    lanes1 = (inside(0, 1) & inside(1, 1) & inside(2, 1)).mean()
    print("frame %3d rows inside +-1: %.4f | y +-2: %.4f | x,y +-2: %.4f | x,y,z +-2: %.4f | voxels inside +-1: %.4f | max reach %s" % (
        frame, w1.mean(), w2y.mean(), w2xy.mean(), w2.mean(), lanes1, [round(float(np.abs(d[a]).max()), 2) for a in range(3)]), flush=True)
