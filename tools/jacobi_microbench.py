#!/usr/bin/env python3
"""Sweep micro-benchmark of SURVEY.md 8d: p = 0, b ~ U(-1,1) from a fixed seed, N lock-step sweeps, no early
out.  Reports microseconds per sweep, cell-updates/s and algorithmic GB/s (12 B per cell-sweep).

    python tools/jacobi_microbench.py [--grid 256] [--iters 40] [--reps 20] [--fuse T]
    python tools/jacobi_microbench.py --sweep          # subprocess per tile configuration (env knobs)
"""
import argparse
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run_once(args):
    import numpy as np
    import fluidx12_amd as fx
    G = args.grid
    Z = args.depth or G
    f = fx.Fluid()
    assert f.Init(800, 800, (G, G, Z), jacobi_iters=args.iters, jacobi_fuse=args.fuse)
    rng = np.random.default_rng(12345)
    f.upload(fx.FIELD_DIVERGENCE, rng.uniform(-1, 1, (Z, G, G)).astype(np.float32))
    f.upload(fx.FIELD_PRESSURE, np.zeros((Z, G, G), np.float32))
    for _ in range(3):
        f.Jacobi(args.iters)
    f.Synchronize()
    f.timing_enable(True)
    f.timing_read(True)
    for _ in range(args.reps):
        f.Jacobi(args.iters)
    f.Synchronize()
    t = f.timing_read(True)
    cells = float(G) * G * Z
    us_per_sweep = t.jacobi_ms * 1e3 / t.jacobi_sweeps
    out = {"grid": [G, G, Z], "iters": args.iters, "fuse": args.fuse,
           "us_per_sweep": us_per_sweep, "launches": int(t.jacobi_launches), "sweeps": int(t.jacobi_sweeps),
           "Gcell_updates_per_s": cells / us_per_sweep / 1e3, "algorithmic_GBps": 12.0 * cells / us_per_sweep / 1e3}
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--depth", type=int, default=0)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--fuse", type=int, default=0)
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--configs", default="")
    args = ap.parse_args()
    if not args.sweep:
        return run_once(args)
    # one run per fused-sweep count (1 = one launch per sweep, 2 / 3 = the register-strip or block kernels, 0 = the default schedule)
    for T in [c for c in args.configs.split(";") if c] or ["1", "2", "3", "0"]:
        r = subprocess.run([sys.executable, __file__, "--grid", str(args.grid), "--iters", str(args.iters), "--reps",
                            str(args.reps), "--fuse", T] + (["--depth", str(args.depth)] if args.depth else []),
                           capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if line:
            d = json.loads(line[-1])
            print("fuse=%s %7.2f us/sweep  %7.1f Gupd/s  %7.0f GB/s" % (T, d["us_per_sweep"], d["Gcell_updates_per_s"], d["algorithmic_GBps"]), flush=True)
        else:
            print("fuse=%s FAILED: %s" % (T, r.stderr[-300:]), flush=True)


if __name__ == "__main__":
    main()
