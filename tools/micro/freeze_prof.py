"""(debug build only) per-phase cycle counts of k_freeze_tiles, accumulated by thread 0 of every workgroup"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fluidx12_amd as fx
from fluidx12_amd import capi
lib = capi.lib() if hasattr(capi, "lib") else ctypes.CDLL(os.path.join(os.path.dirname(fx.__file__), "libfluidx_hip.so"))
buf = (ctypes.c_ulonglong * 16)()
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 128
f = fx.Fluid(); assert f.Init(0, 0, (grid,) * 3, storage="fp16", jacobi_iters=64, jacobi_mode="faithful")
dt = np.float32(f.default_time_step())
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for k in range(warm): f.UpdateFrame(dt, k % 3); f.Simulate(k % 3)
f.Synchronize(); lib.fx_debug_freeze_prof(buf, 1)
passes = (ctypes.c_uint * 160)(); lib.fx_debug_freeze_passes(passes, 1)
n = 20
for k in range(n): f.UpdateFrame(dt, k % 3); f.Simulate(k % 3)
f.Synchronize(); lib.fx_debug_freeze_prof(buf, 0); lib.fx_debug_freeze_passes(passes, 0)
v = list(buf); tiles = max(v[15], 1)
names = ["to tile start", "barrier (LDS free)", "loads landed", "LDS stored+barrier", "level 1", "level 2", "level 3", "level 4", "core stored", "appended"]
print("grid", grid, "warm", warm, "tile passes per step", tiles / n)
print("loads issued (address math)  %9.1f" % (v[10] / tiles))
for i, nm in enumerate(names): print("%-22s %9.1f ticks per tile pass" % (nm, v[i] / tiles))
print("sum", sum(v[:11]) / tiles)
pp = list(passes)
print("tile passes per launch (first level: passes / copy-only):", " ".join("%d:%.0f/%.0f" % (l, pp[l] / n, pp[80 + l] / n) for l in range(80) if pp[l] or pp[80 + l]))
