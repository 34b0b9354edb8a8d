#!/usr/bin/env python3
"""Long-run check: the default schedule (multi-sweep register / LDS kernels, LDS-staged advection, four-cell projection) against
the plainest kernels the library has (one Jacobi sweep per launch in k_jacobi_v4 / k_jacobi_generic, gather advection, scalar
projection / divergence) over hundreds of steps from the zero state -- every field must end bit-identical.

    python tools/long_run_parity.py [grid] [steps] [storage] [faithful]

With `faithful`: the reference's own configuration (<= 64 sweeps + early-out) -- the sparse solver, fused divergence included, against
k_jacobi_generic with the byte mask (FLUIDX_FREEZE_FAST=0).
"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
storage = sys.argv[3] if len(sys.argv) > 3 else "fp32"
faithful = len(sys.argv) > 4 and sys.argv[4] == "faithful"
code = (
    "import sys, hashlib, numpy as np\n"
    "sys.path.insert(0, %r)\n"
    "import fluidx12_amd as fx\n"
    "f = fx.Fluid(); assert f.Init(64, 64, (%d, %d, %d), jacobi_iters=%d, jacobi_mode=%r, storage=%r, **({'jacobi_fuse': 1} if %s else {}))\n"
    "dt = np.float32(f.default_time_step())\n"
    "for k in range(%d):\n"
    "    f.UpdateFrame(dt, k %% 3); f.Simulate(k %% 3)\n"
    "    if k %% 100 == 99 or k == %d - 1:\n"
    "        f.Synchronize(); h = hashlib.sha256()\n"
    "        for fid in (fx.FIELD_VELOCITY, fx.FIELD_COLOR, fx.FIELD_PRESSURE):\n"
    "            a = f.download(fid); assert np.isfinite(a).all(); h.update(a.tobytes())\n"
    "        print('DIGEST', k + 1, h.hexdigest())\n")
plain = dict(FLUIDX_ADVECT_LDS="0", FLUIDX_ADVECT_FAST="0", FLUIDX_PROJECT_V4="0", FLUIDX_ROW_VW="0", FLUIDX_JACOBI_BLOCK="0", FLUIDX_JACOBI_BLOCKG="0")
if faithful:
    plain["FLUIDX_FREEZE_FAST"] = "0"
out = []
for simple in (False, True):
    env = dict(os.environ, **(plain if simple else {}))
    r = subprocess.run([sys.executable, "-c", code % (ROOT, grid, grid, grid, 64 if faithful else 40, "faithful" if faithful else "fixed", storage, simple and not faithful, steps, steps)], env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out.append([l for l in r.stdout.splitlines() if l.startswith("DIGEST")])
for a, b in zip(*out):
    print(a, "==" if a == b else "!=", b.split()[-1][:12])
sys.exit(0 if out[0] == out[1] and out[0] else 1)
