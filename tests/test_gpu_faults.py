"""A failure inside a kernel must be LOUD.  The strip kernels' LDS hand-overs wait in bounded loops (65 536 polls, ~10 ms; a neighbour is
normally a z step -- ~2 us -- away); a wait that runs out raises a device word which fx_synchronize returns as FX_E_DEVICE (round 5
continued silently with whatever the wave had read: a wrong pressure field instead of a hang; the three-sweep kernels span unbounded).

Forced here with a lab variant of the library (tools/_variants/libfluidx_hip_fault.so = fx_jacobi_strip4.hip compiled with
-DFX_LAB_DROP_PUBLISH: one wave of every workgroup never posts its level-2 counter), loaded into a child process through
FLUIDX_LIB_PATH.  The shipped library runs the same calls clean."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import numpy as np, sys
import fluidx12_amd as fx
from fluidx12_amd import capi
f = fx.Fluid()
assert f.Init(800, 800, (256, 256, 24), jacobi_iters=4, jacobi_fuse=4)
rng = np.random.default_rng(3)
f.upload(fx.FIELD_PRESSURE, rng.standard_normal((24, 256, 256)).astype(np.float32))
f.upload(fx.FIELD_DIVERGENCE, rng.uniform(-1, 1, (24, 256, 256)).astype(np.float32))
f.Jacobi(4)
try:
    f.Synchronize()
    print("SYNC OK")
except fx.FluidxError as e:
    print("SYNC ERROR", e.status, str(e))
f.Synchronize()                      # the word was taken with the report: the context is usable again
print("SECOND SYNC OK")
"""


def fault_lib():
    from fluidx12_amd import build
    lib = os.path.join(ROOT, "tools", "_variants", "libfluidx_hip_fault.so")
    srcs = [os.path.join(build.CSRC, "fx_jacobi_strip4.hip"), os.path.join(build.CSRC, "fx_context.cpp"), os.path.join(build.CSRC, "fx_internal.h")]
    if not os.path.exists(lib) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in srcs):
        build.build_variant("fault", ["-DFX_LAB_DROP_PUBLISH"])
    return lib


def run_child(env_extra):
    env = dict(os.environ, **env_extra)
    return subprocess.run([sys.executable, "-c", CHILD], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)


def test_a_hand_over_wait_that_runs_out_is_reported_by_fx_synchronize():
    r = run_child({"FLUIDX_LIB_PATH": fault_lib()})
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    assert "SYNC ERROR -2" in r.stdout and "hand-over wait ran out" in r.stdout, r.stdout       # FX_E_DEVICE, and it says which kernel family
    assert "SECOND SYNC OK" in r.stdout


def test_the_shipped_library_runs_the_same_calls_clean():
    r = run_child({})
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    assert "SYNC OK" in r.stdout and "SYNC ERROR" not in r.stdout
