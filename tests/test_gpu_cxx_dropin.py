"""The C++ drop-in executed on the device: examples/fluidx_demo.cpp -- the reference's frame loop (FluidX12.cpp:257-284, 435-588) written
against csrc/Fluid.hpp, the mirror of class Fluid (Content/Fluid.h:20-35) -- is built with hipcc against libfluidx_hip.so, run as a
child process in the reference's configuration (RGBA16F fields, ITER = 64 with the per-cell early-out, CLAMP like FluidEZ), and
what it leaves behind (state file and screen shot) is compared with the same frames driven through the Python mirror."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import fluidx12_amd as fx
from fluidx12_amd import build as fxbuild

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


def build_demo(tmp_path):
    cc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(cc):
        pytest.skip("no hipcc on this box")
    lib = fxbuild.ensure_built()
    exe = str(tmp_path / "fluidx_demo")
    libdir = os.path.dirname(lib)
    subprocess.run([cc, "-std=c++17", "-O2", os.path.join(ROOT, "examples", "fluidx_demo.cpp"), "-o", exe, "-L" + libdir, "-lfluidx_hip",
                    "-Wl,-rpath," + libdir], check=True, timeout=600)
    return exe


@pytest.mark.parametrize("grid,frames", [((64, 64, 64), 8), ((48, 48, 40), 5)])
def test_cxx_demo_leaves_the_python_path_s_fields_and_picture(tmp_path, grid, frames):
    exe = build_demo(tmp_path)
    ppm, ck = str(tmp_path / "shot.ppm"), str(tmp_path / "state.fxck")
    r = subprocess.run([exe, "-gridSize", *(str(v) for v in grid), "-frames", str(frames), "-screenshot", ppm, "-checkpoint", ck],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "%d frames of %dx%dx%d" % ((frames,) + grid) in r.stdout
    state = fx.read_checkpoint(ck)
    assert state["grid"] == grid and state["steps"] == frames and state["complete"].all()
    with open(ppm, "rb") as fp:
        assert fp.readline() == b"P6\n" and fp.readline() == b"800 800\n" and fp.readline() == b"255\n"
        shot = np.frombuffer(fp.read(), np.uint8).reshape(800, 800, 3)

    # the same run saved as a PNG (the reference's screen-shot format, FluidX12.cpp:640-660): a valid file -- signature, chunk CRCs, a zlib
    # stream any inflater takes -- holding exactly the PPM's pixels
    if frames == 5:
        import struct
        import zlib
        png = str(tmp_path / "shot.png")
        r = subprocess.run([exe, "-gridSize", *(str(v) for v in grid), "-frames", str(frames), "-screenshot", png], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        raw = open(png, "rb").read()
        assert raw[:8] == b"\x89PNG\r\n\x1a\n"
        pos, chunks = 8, []
        while pos < len(raw):
            n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
            body = raw[pos + 8:pos + 8 + n]
            assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + body)
            chunks.append((typ, body))
            pos += 12 + n
        assert [t for t, _ in chunks] == [b"IHDR", b"IDAT", b"IEND"]
        assert struct.unpack(">IIBBBBB", chunks[0][1]) == (800, 800, 8, 6, 0, 0, 0)
        rows = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(800, 1 + 4 * 800)
        assert not rows[:, 0].any()                                   # filter type 0 on every row
        assert np.array_equal(rows[:, 1:].reshape(800, 800, 4)[..., :3], shot)

    # the same frames through the Python mirror of the operator surface (what every other GPU test drives)
    f = fx.Fluid()
    assert f.Init(800, 800, grid, storage="fp16", jacobi_iters=64, jacobi_mode="faithful", advect_address="clamp")
    f.SetMaxSamples(192, 64)
    view, proj, eye = fx.default_camera(800, 800)
    dt = f32(2.0) / f32(grid[1])                                     # FluidX12.cpp:266
    for k in range(frames):
        f.UpdateFrame(dt, k % 3, view, proj, eye)
        f.Simulate(k % 3)
        f.ClearRenderTarget((0.2, 0.2, 0.2, 0.0))
        f.Render(k % 3, fx.Fluid.OPTIMIZED, to_target=True)
    f.Synchronize()
    assert np.array_equal(f.download(fx.FIELD_VELOCITY).view(np.uint32), state["velocity"].view(np.uint32))
    assert np.array_equal(f.download(fx.FIELD_COLOR).view(np.uint32), state["color"].view(np.uint32))
    assert np.array_equal(f.download(fx.FIELD_PRESSURE).view(np.uint32), state["pressure"].view(np.uint32))
    assert state["color"][..., 3].max() > 0.2                       # there is smoke
    img = f.download(fx.FIELD_TARGET)
    # the two hosts build the camera matrices on their own (libm there, numpy here): the pictures may part by a rounding of a ray
    d = np.abs(img[..., :3].astype(np.int32) - shot.astype(np.int32))
    assert d.max() <= 2 and (d > 0).mean() < 0.01, (int(d.max()), float((d > 0).mean()))
    assert (shot != 51).any()                                       # the volume shows on the cleared target
