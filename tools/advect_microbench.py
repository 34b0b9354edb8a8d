#!/usr/bin/env python3
"""Time the advection kernel alone on the state the bench produces, under several kernel selections, and check that every
selection writes the same bits.

    python tools/advect_microbench.py [--grid 256] [--steps 25] [--reps 20] [--iters 40]

Prints one JSON line per variant: {variant, ms per launch (HIP events around the launch), algorithmic GB/s = (2V + 2C) x voxels / time,
identical: outputs equal the reference variant's bit for bit}.  FLUIDX_ADVECT_* are read per launch, so one process covers them all."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluidx12_amd as fx   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--storage", default="fp32")
    ap.add_argument("--variants", default="fast;lds;lds,zchunk=8;lds,zchunk=32;lds,zchunk=64")
    a = ap.parse_args()
    G = a.grid
    f = fx.Fluid()
    assert f.Init(800, 800, (G, G, G), jacobi_iters=a.iters, storage=a.storage)
    dt = np.float32(f.default_time_step())
    from fluidx12_amd import capi
    capi.set_knob("ADVECT_LDS", "0")
    for k in range(a.steps):
        f.UpdateFrame(dt, k % 3)
        f.Simulate(k % 3)
    f.UpdateFrame(dt, 0)
    f.Synchronize()
    bpv = 56 if a.storage == "fp32" else 28
    ref = None
    for var in a.variants.split(";"):
        opts = var.split(",")
        capi.set_knob("ADVECT_LDS", "2" if opts[0] == "lds" else "0")
        capi.set_knob("ADVECT_FAST", "0" if opts[0] == "generic" else "1")
        capi.set_knob("ADVECT_ZCHUNK", None)
        for o in opts[1:]:
            k, v = o.split("=")
            capi.set_knob("ADVECT_" + k.upper(), v)
        f.Advect(); f.Synchronize()
        f.timing_enable(True); f.timing_read(True)
        for _ in range(a.reps):
            f.Advect()
        f.Synchronize()
        t = f.timing_read(True)
        f.timing_enable(False)
        out = (f.download(fx.FIELD_VELOCITY1), f.download(fx.FIELD_COLOR))
        if ref is None:
            ref = out
        same = bool(np.array_equal(out[0].view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(out[1].view(np.uint32), ref[1].view(np.uint32)))
        ms = t.advect_ms / a.reps
        print(json.dumps({"variant": var, "grid": G, "state_step": a.steps, "ms": round(ms, 5),
                          "algorithmic_GBps": round(bpv * float(G) ** 3 / (ms * 1e-3) / 1e9, 1), "identical": same}), flush=True)


if __name__ == "__main__":
    main()
