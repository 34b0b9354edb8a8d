"""Times the reference-configuration step (FX_JACOBI_FAITHFUL: 64-sweep cap + early-out, RGBA16F storage) with the sparse solver of
fx_jacobi_freeze.hip against the one-sweep-per-launch kernel, and its launch-shape knobs.  GPU box only.

    python tools/freeze_bench.py [--grid 256] [--warm 40] [--steps 40] [--storage fp16] [--variants]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluidx12_amd as fx   # noqa: E402


def run(grid, warm, steps, storage, env):
    from fluidx12_amd import capi
    for k, v in env.items():                                  # launcher switches (fx_set_knob), named FLUIDX_<NAME> on the command line
        capi.set_knob(k[len("FLUIDX_"):], v)
    try:
        f = fx.Fluid()
        assert f.Init(0, 0, (grid, grid, grid), storage=storage, jacobi_iters=64, jacobi_mode="faithful")
        dt = np.float32(f.default_time_step())
        for k in range(warm):
            f.UpdateFrame(dt, k % 3); f.Simulate(k % 3)
        f.Synchronize()
        f.timing_read(True)
        t0 = time.perf_counter()
        for k in range(steps):
            f.UpdateFrame(dt, k % 3); f.Simulate(k % 3)
        f.Synchronize()
        wall = (time.perf_counter() - t0) / steps * 1e3
        sw = f.timing_read(True)
        # stage times on a few marked steps
        f.timing_enable(True)
        for k in range(8):
            f.UpdateFrame(dt, k % 3); f.Simulate(k % 3)
        f.Synchronize()
        t = f.timing_read(True)
        f.timing_enable(False)
        f.Release()
        n = max(t.steps, 1)
        return dict(env=env, ms_per_step=round(wall, 4), voxel_updates_per_s=round(grid ** 3 / wall * 1e3 / 1e9, 3),
                    sweeps_per_solve=round(sw.freeze_sweeps / max(sw.freeze_solves, 1), 2) if sw.freeze_solves else None,
                    advect_ms=round(t.advect_ms / n, 4), divergence_ms=round(t.divergence_ms / n, 4), jacobi_ms=round(t.jacobi_ms / n, 4),
                    project_ms=round(t.project_ms / n, 4), jacobi_launches=t.jacobi_launches // n)
    finally:
        for k in env:
            capi.set_knob(k[len("FLUIDX_"):], None)


def all_active(grid, levels, env):
    """every tile listed, no cell ever freezing: what one tile launch of `levels` levels costs per tile"""
    from fluidx12_amd import capi
    for k, v in env.items():                                  # launcher switches (fx_set_knob), named FLUIDX_<NAME> on the command line
        capi.set_knob(k[len("FLUIDX_"):], v)
    try:
        f = fx.Fluid()
        assert f.Init(0, 0, (grid, grid, grid), storage="fp32", jacobi_iters=1 + levels, jacobi_mode="faithful")
        rng = np.random.default_rng(1)
        p = (rng.standard_normal((grid, grid, grid)) * 50).astype(np.float32)
        f.upload(fx.FIELD_DIVERGENCE, p)
        out = []
        for rep in range(6):
            f.upload(fx.FIELD_PRESSURE, p)
            f.timing_enable(True); f.timing_read(True)
            f.Jacobi(1 + levels)
            f.Synchronize()
            t = f.timing_read(True)
            out.append((round(t.jacobi_main_ms * 1e3, 2), round((t.jacobi_ms - t.jacobi_main_ms) * 1e3, 2), t.freeze_sweeps))
        f.Release()
        return dict(env=env, grid=grid, levels=levels, dense_us_tiles_us_sweeps=out[2:])
    finally:
        for k in env:
            capi.set_knob(k[len("FLUIDX_"):], None)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--warm", type=int, default=40)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--storage", default="fp16")
    ap.add_argument("--variants", action="store_true")
    ap.add_argument("--all-active", action="store_true", help="one tile launch over EVERY tile (nothing freezes): the per-tile cost")
    a = ap.parse_args()
    if a.all_active:
        for t in (4, 2, 1):
            for nt in (512, 256, 1024):
                print(json.dumps(all_active(a.grid, t, {"FLUIDX_FREEZE_T": t, "FLUIDX_FREEZE_NT": nt})), flush=True)
        sys.exit(0)
    cases = [{"FLUIDX_FREEZE_FAST": 1}, {"FLUIDX_FREEZE_FAST": 0}]
    if a.variants:
        cases += [{"FLUIDX_FREEZE_T": t, "FLUIDX_FREEZE_NT": nt, "FLUIDX_FREEZE_WGS": w} for t in (2, 3, 4) for nt in (512, 1024) for w in (512, 1024, 2048)]
    for env in cases:
        print(json.dumps(dict(grid=a.grid, storage=a.storage, **run(a.grid, a.warm, a.steps, a.storage, env))), flush=True)
