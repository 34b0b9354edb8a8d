"""Render-leg micro-benchmark: the marches of one developed plume, accelerated (FX_OPT_RENDER_ACCEL 1) and plain (0), timed by the
library's own HIP-event marks.  Run on the GPU box (optionally under tools/kstats.sh for per-kernel times):

    python3 tools/render_bench.py [--grid 256] [--frame 132] [--storage fp32] [--sh] [--reps 5] [--modes 1,0] [--flags optimized,merged,direct]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluidx12_amd as fx                      # noqa: E402
from fluidx12_amd import capi                  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--frame", type=int, default=132)
    ap.add_argument("--storage", default="fp32")
    ap.add_argument("--sh", action="store_true")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--modes", default="1,0")
    ap.add_argument("--flags", default="optimized")
    ap.add_argument("--viewport", default="1920x1080")
    a = ap.parse_args()
    W, H = (int(v) for v in a.viewport.split("x"))
    G = a.grid
    f = fx.Fluid()
    assert f.Init(W, H, (G, G, G), storage=a.storage)
    view, proj, eye = fx.default_camera(W, H)
    for k in range(a.frame):
        f.UpdateFrame(np.float32(f.default_time_step()), k % 3, view, proj, eye)
        f.Simulate(k % 3)
        if k == a.frame - 2:
            f.Render(k % 3, fx.Fluid.OPTIMIZED)   # a context that renders its frames: the last step's advection writes the render's side volume (bench.py does the same)
    if a.sh:
        sh = (np.random.default_rng(5).random((9, 3)) * np.array([[2.0]] + [[0.5]] * 8)).astype(np.float32)
        f.SetSH(sh)
    f.UpdateFrame(0.0, 0, view, proj, eye)
    f.Synchronize()
    flagmap = {"optimized": fx.Fluid.OPTIMIZED, "merged": fx.Fluid.RAY_MARCH_CUBEMAP, "direct": fx.Fluid.SEPARATE_LIGHT_PASS,
               "direct_merged": fx.Fluid.RAY_MARCH_DIRECT}
    out = {"grid": G, "frame": a.frame, "storage": a.storage, "sh": a.sh, "runs": []}
    pics = {}
    for mode in (int(m) for m in a.modes.split(",")):
        f.set_option(capi.OPT_RENDER_ACCEL, mode)
        for name in a.flags.split(","):
            fl = flagmap[name]
            f.ClearRenderTarget()
            f.Render(0, fl)
            f.Synchronize()
            key = (name,)
            pic = (f.download(fx.FIELD_LIGHTMAP).tobytes() if fl & fx.Fluid.SEPARATE_LIGHT_PASS else b"") + \
                (f.download(fx.FIELD_CUBEMAP).tobytes() if fl & fx.Fluid.RAY_MARCH_CUBEMAP else f.download(fx.FIELD_TARGET_FLOAT).tobytes())
            same = None
            if key in pics:
                same = pics[key] == pic
            pics.setdefault(key, pic)
            f.timing_enable(True)
            f.timing_read(reset=True)
            for _ in range(a.reps):
                f.Render(0, fl)
            f.Synchronize()
            t = f.timing_read(reset=True)
            f.timing_enable(False)
            out["runs"].append({"accel": mode, "flags": name, "light_ms": t.light_ms / a.reps, "view_ms": t.view_ms / a.reps,
                                "same_picture_as_first_mode": same})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
