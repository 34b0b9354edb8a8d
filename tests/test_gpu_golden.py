"""The HIP kernels (through the C ABI) against golden vectors produced by the reference's OWN shipped shader
binaries (tests/golden/dxbc_*.npz; see tests/test_dxbc_golden.py and tools/make_dxbc_golden.py).
This is the product checked against the reference itself, with no oracle in between."""
import os

import numpy as np
import pytest

import fluidx12_amd as fx

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SIM = np.load(os.path.join(GOLD, "dxbc_sim.npz"))
REN = np.load(os.path.join(GOLD, "dxbc_render.npz"))
SH = np.load(os.path.join(GOLD, "dxbc_sh.npz"))
RES = np.load(os.path.join(GOLD, "dxbc_resolve.npz"))
DIR = np.load(os.path.join(GOLD, "dxbc_direct.npz"))
f32 = np.float32


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    n = np.sqrt((b ** 2).sum())
    return np.sqrt(((a - b) ** 2).sum()) / (n if n > 0 else 1.0)


def make(dims, **kw):
    f = fx.Fluid()
    assert f.Init(640, 480, dims, **kw), f.last_status
    return f


@pytest.mark.parametrize("tag,dims", [("3d", (16, 16, 8)), ("2d", (16, 16, 1))])
@pytest.mark.parametrize("fmt", ["f32", "f16"])
@pytest.mark.parametrize("address", ["clamp", "mirror"])
def test_advect_vs_reference_binary(tag, dims, fmt, address):
    k = "advect_%s_%s_%s" % (tag, fmt, address)
    f = make(dims, storage="fp16" if fmt == "f16" else "fp32", advect_address=address)
    f.upload(fx.FIELD_VELOCITY, SIM[k + "_vel_in"])
    f.upload(fx.FIELD_COLOR, SIM[k + "_col_in"])
    f.UpdateFrame(f32(f.default_time_step()), 0)
    f.Advect()
    f.Synchronize()
    gv, gc = f.download(fx.FIELD_VELOCITY1), f.download(fx.FIELD_COLOR)
    # one exp2 per voxel (device v_exp_f32 vs correctly rounded): everything else is bit-exact
    assert rel_l2(gv, SIM[k + "_vel_out"]) < 1e-6 and rel_l2(gc, SIM[k + "_col_out"]) < 1e-6
    assert np.mean(gv != SIM[k + "_vel_out"]) < 2e-3 and np.mean(gc != SIM[k + "_col_out"]) < 2e-3


@pytest.mark.parametrize("tag,dims", [("3d", (16, 16, 8)), ("2d", (16, 16, 1))])
@pytest.mark.parametrize("fmt", ["f32", "f16"])
def test_project_vs_reference_binary(tag, dims, fmt):
    """CSProject3D/2D.cso (lock-step schedule) == divergence + faithful 64-sweep Jacobi + projection kernels, bit-exact"""
    k = "project_%s_%s" % (tag, fmt)
    f = make(dims, storage="fp16" if fmt == "f16" else "fp32", jacobi_iters=64, jacobi_mode="faithful")
    f.upload(fx.FIELD_VELOCITY1, SIM[k + "_vel_in"])
    f.upload(fx.FIELD_PRESSURE, SIM[k + "_p_in"])
    f.UpdateFrame(f32(f.default_time_step()), 0)
    f.Divergence()
    f.Jacobi(64)
    f.Project()
    f.Synchronize()
    assert np.array_equal(f.download(fx.FIELD_PRESSURE), SIM[k + "_p_out"])
    assert np.array_equal(f.download(fx.FIELD_VELOCITY), SIM[k + "_vel_out"])


@pytest.mark.parametrize("tag,dims,address", [("3d", (16, 16, 16), "mirror"), ("3d_ez", (16, 16, 16), "clamp"), ("2d", (32, 32, 1), "clamp")])
def test_rollout_vs_reference_binaries(tag, dims, address):
    """4 frames in the reference's own configuration: RGBA16F fields, ITER 64 + early-out, Fluid (MIRROR) / FluidEZ (CLAMP)"""
    f = make(dims, storage="fp16", jacobi_iters=64, jacobi_mode="faithful", advect_address=address)
    for st in range(1, 5):
        f.UpdateFrame(f32(f.default_time_step()), (st - 1) % 3)
        f.Simulate((st - 1) % 3)
        f.Synchronize()
        for field, key in ((fx.FIELD_VELOCITY, "vel"), (fx.FIELD_COLOR, "col"), (fx.FIELD_PRESSURE, "p")):
            got, ref = f.download(field), SIM["rollout_%s_step%d_%s" % (tag, st, key)]
            assert rel_l2(got, ref) < 1e-4, (st, key)                  # north_star tolerance
            assert np.mean(got != ref) < 5e-3, (st, key)               # in practice (almost) every value is identical


@pytest.mark.parametrize("has_sh", [0, 1])
def test_ray_march_vs_reference_binaries(has_sh):
    X, S, nl, ns, nml, mask, vw, vh = (int(v) for v in REN["params"])
    col = REN["color"]
    f = make((X, X, X), storage="fp16")                                # the golden volume is RGBA16F-exact
    view, proj, eye = fx.default_camera(vw, vh)
    f.upload(fx.FIELD_COLOR, col)
    if has_sh:
        f.SetSH(REN["sh"])

    def cube_close(got, ref):
        d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 0.01, (int(d.max()), float((d > 0).mean()))

    f.SetMaxSamples(ns, nl)
    f.UpdateFrame(0.0, 0, view, proj, eye)
    fi = f.frame_info()
    assert (fi.cube_lod, fi.ray_samples, fi.visibility_mask, fi.cube_size) == (0, ns, mask, S)
    f.Render(0, fx.Fluid.OPTIMIZED)
    f.Synchronize()
    lm, ref = f.download(fx.FIELD_LIGHTMAP), REN["lightmap_sh%d" % has_sh]
    assert np.mean(lm != ref) < 2e-3 and np.abs(lm - ref).max() <= np.abs(ref).max() * 2.0 ** -5     # CSRayMarchL.cso
    cube_close(f.download(fx.FIELD_CUBEMAP), REN["cube_separate_sh%d" % has_sh])                     # CSRayMarchV.cso
    f.SetMaxSamples(ns, nml)
    f.UpdateFrame(0.0, 0, view, proj, eye)
    f.Render(0, fx.Fluid.RAY_MARCH_CUBEMAP)
    f.Synchronize()
    cube_close(f.download(fx.FIELD_CUBEMAP), REN["cube_merged_sh%d" % has_sh])                       # CSRayMarch.cso


@pytest.mark.parametrize("name", ["rendered16", "random8"])
def test_cube_resolve_vs_reference_binary(name):
    """k_resolve_cube against PSRayCastCube.cso's own output (row f-1).  The library derives its frame constants itself
    (fp32 DirectXMath restatement, the golden ones were rounded from float64), so a silhouette pixel may flip and values
    agree to ~1e-5; tests/test_gpu_render.py holds the bit-exact comparison with identical constants."""
    W, H, vw, vh = (int(v) for v in RES["params"])
    cube = RES["cube_" + name]
    N = cube.shape[1]
    f = fx.Fluid()
    assert f.Init(W, H, (N, N, N))
    view, proj, eye = fx.default_camera(vw, vh)
    f.UpdateFrame(0.0, 0, view, proj, eye)
    assert f.frame_info().cube_size == N
    f.upload(fx.FIELD_CUBEMAP, cube)
    f.ClearRenderTarget()
    f.RenderCube(0)
    f.Synchronize()
    got = f.download(fx.FIELD_TARGET_FLOAT)
    ref, disc = RES["target_" + name], RES["discard_" + name]
    cov = got[..., 3] > 0
    assert np.mean(cov != ~disc) < 2e-3
    both = cov & ~disc
    d = np.abs(got[both] - ref[both])
    assert np.mean(d > 1e-4) < 2e-3 and np.median(d) < 2e-5


@pytest.mark.parametrize("name", ["rendered16", "random8"])
def test_cube_resolve_vs_the_rasterised_resolve_as_shipped(name):
    """k_resolve_cube against the picture the reference's executable DRAWS (VSCube.cso + PSCube.cso through D3D11's rasteriser rules:
    tests/golden/dxbc_raster.npz, tools/make_raster_golden.py): the same pixels up to the silhouette flips of the library's own fp32
    frame constants, colours within half an 8-bit step"""
    RAS = np.load(os.path.join(GOLD, "dxbc_raster.npz"))
    W, H, vw, vh = (int(v) for v in RES["params"])
    cube = RES["cube_" + name]
    N = cube.shape[1]
    f = fx.Fluid()
    assert f.Init(W, H, (N, N, N))
    view, proj, eye = fx.default_camera(vw, vh)
    f.UpdateFrame(0.0, 0, view, proj, eye)
    f.upload(fx.FIELD_CUBEMAP, cube)
    f.ClearRenderTarget()
    f.RenderCube(0)
    f.Synchronize()
    got = f.download(fx.FIELD_TARGET_FLOAT)
    drawn = ~RAS["discard_" + name]
    cov = got[..., 3] > 0
    assert np.mean(cov != drawn) < 2e-3
    both = cov & drawn
    d = np.abs(got[both] - RAS["target_" + name][both])
    assert np.mean(d.max(axis=-1) > 1 / 510) < 2e-3 and d.max() < 1 / 255


@pytest.mark.parametrize("has_sh", [0, 1])
def test_direct_ray_cast_vs_reference_binaries(has_sh):
    """k_raycast_direct against PSRayCastV.cso / PSRayCast.cso (row f-2); own frame constants as in the resolve test"""
    W, H, ns, nml, vw, vh = (int(v) for v in DIR["params"])
    X = int(REN["params"][0])
    f = fx.Fluid()
    assert f.Init(W, H, (X, X, X), storage="fp16")
    view, proj, eye = fx.default_camera(vw, vh)
    f.upload(fx.FIELD_COLOR, REN["color"])
    if has_sh:
        f.SetSH(REN["sh"])

    def close(key):
        got = f.download(fx.FIELD_TARGET_FLOAT)
        ref = DIR[key]                                      # zeros where discarded or where the ray met no smoke
        lit = (ref[..., 3] > 0) | (got[..., 3] > 0)
        assert lit.mean() > 0.1
        d = np.abs(got[lit] - ref[lit])
        assert np.mean(d > 2e-3) < 5e-3 and np.median(d) < 1e-4, (float(np.mean(d > 2e-3)), float(np.median(d)))

    f.SetMaxSamples(ns, int(REN["params"][2]))
    f.UpdateFrame(0.0, 0, view, proj, eye)
    assert f.frame_info().ray_samples == ns
    f.ClearRenderTarget()
    f.Render(0, fx.Fluid.SEPARATE_LIGHT_PASS)               # rayMarchL + rayCastVDirect
    f.Synchronize()
    close("separate_sh%d" % has_sh)
    f.SetMaxSamples(ns, nml)
    f.UpdateFrame(0.0, 0, view, proj, eye)
    f.ClearRenderTarget()
    f.Render(0, fx.Fluid.RAY_MARCH_DIRECT)                  # rayCastDirect
    f.Synchronize()
    close("merged_sh%d" % has_sh)


def test_2d_visualiser_vs_reference_binary():
    col = DIR["visualize_color"]
    f = fx.Fluid()
    assert f.Init(40, 30, (24, 24, 1), storage="fp16")
    f.upload(fx.FIELD_COLOR, col)
    f.UpdateFrame(0.0, 0)
    f.Render(0, 0)
    f.Synchronize()
    assert np.array_equal(f.download(fx.FIELD_TARGET_FLOAT).view(np.uint32), DIR["visualize_target"].view(np.uint32))


def test_sh_transform_vs_reference_binaries():
    f = make((16, 16, 16))
    lp = fx.LightProbe(f)
    assert lp.Init(SH["cube"])
    lp.TransformSH()
    assert np.allclose(lp.GetSH(), SH["sh"], rtol=1e-5, atol=1e-6)


# ---- round 5: tests/golden/dxbc_wide.npz (over-covered grids, an 8-frame rollout, cube LOD 1, the SH chain at 256^2) ----------------
WIDE = np.load(os.path.join(GOLD, "dxbc_wide.npz"))


@pytest.mark.parametrize("tag,dims", [("3d", (20, 20, 10)), ("2d", (12, 12, 1))])
@pytest.mark.parametrize("address", ["clamp", "mirror"])
def test_advect_on_over_covered_grids_vs_reference_binary(tag, dims, address):
    """grids that are no multiple of the reference's 8 x 8 thread group (its surplus threads load zeros and drop their stores)"""
    k = "advect_%s_%s" % (tag, address)
    f = make(dims, advect_address=address)
    f.upload(fx.FIELD_VELOCITY, WIDE[k + "_vel_in"])
    f.upload(fx.FIELD_COLOR, WIDE[k + "_col_in"])
    f.UpdateFrame(f32(f.default_time_step()), 0)
    f.Advect()
    f.Synchronize()
    gv, gc = f.download(fx.FIELD_VELOCITY1), f.download(fx.FIELD_COLOR)
    assert rel_l2(gv, WIDE[k + "_vel_out"]) < 1e-6 and rel_l2(gc, WIDE[k + "_col_out"]) < 1e-6
    assert np.mean(gv != WIDE[k + "_vel_out"]) < 2e-3 and np.mean(gc != WIDE[k + "_col_out"]) < 2e-3


@pytest.mark.parametrize("tag,dims", [("3d", (20, 20, 10)), ("2d", (12, 12, 1))])
def test_project_on_over_covered_grids_vs_reference_binary(tag, dims):
    k = "project_%s" % tag
    f = make(dims, jacobi_iters=64, jacobi_mode="faithful")
    f.upload(fx.FIELD_VELOCITY1, WIDE[k + "_vel_in"])
    f.upload(fx.FIELD_PRESSURE, WIDE[k + "_p_in"])
    f.UpdateFrame(f32(f.default_time_step()), 0)
    f.Divergence()
    f.Jacobi(64)
    f.Project()
    f.Synchronize()
    assert np.array_equal(f.download(fx.FIELD_PRESSURE), WIDE[k + "_p_out"])
    assert np.array_equal(f.download(fx.FIELD_VELOCITY), WIDE[k + "_vel_out"])


def test_eight_frame_rollout_vs_reference_binaries():
    """8 frames of class Fluid's configuration (MIRROR, RGBA16F, ITER 64 + early-out) at 20 x 20 x 12"""
    f = make((20, 20, 12), storage="fp16", jacobi_iters=64, jacobi_mode="faithful", advect_address="mirror")
    for st in range(1, 9):
        f.UpdateFrame(f32(f.default_time_step()), (st - 1) % 3)
        f.Simulate((st - 1) % 3)
        if st in (2, 5, 8):
            f.Synchronize()
            for field, key in ((fx.FIELD_VELOCITY, "vel"), (fx.FIELD_COLOR, "col"), (fx.FIELD_PRESSURE, "p")):
                got, ref = f.download(field), WIDE["rollout8_step%d_%s" % (st, key)]
                assert rel_l2(got, ref) < 1e-4, (st, key)              # north_star tolerance
                assert np.mean(got != ref) < 1e-2, (st, key)


def test_light_volume_on_an_over_covered_grid_vs_reference_binary():
    """CSRayMarchL.cso on 18^3 voxels (ceil(18 / 4) groups per axis), through fx_render's separate light pass"""
    X = 18
    f = make((X, X, X), storage="fp16")
    view, proj, eye = fx.default_camera(640, 480)
    f.upload(fx.FIELD_COLOR, WIDE["light18_color"])
    f.SetMaxSamples(24, 16)
    f.UpdateFrame(0.0, 0, view, proj, eye)
    f.Render(0, fx.Fluid.OPTIMIZED)
    f.Synchronize()
    lm, ref = f.download(fx.FIELD_LIGHTMAP), WIDE["light18_lightmap"]
    assert np.mean(lm != ref) < 2e-3 and np.abs(lm - ref).max() <= np.abs(ref).max() * 2.0 ** -5


def test_view_march_into_a_coarser_cube_mip_vs_reference_binaries():
    """the 16^3 volume of dxbc_render.npz behind a 20 x 15 viewport: UpdateFrame derives cube LOD 1 (8^2 texels) and 7 samples"""
    X, S, nl, ns, nml, mask, vw, vh = (int(v) for v in WIDE["cube_lod1_params"])
    f = fx.Fluid()
    assert f.Init(vw, vh, (X, X, X), storage="fp16"), f.last_status
    view, proj, eye = fx.default_camera(vw, vh)
    f.upload(fx.FIELD_COLOR, REN["color"])

    def cube_close(got, ref):
        d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 0.02, (int(d.max()), float((d > 0).mean()))

    f.SetMaxSamples(48, nl)
    f.UpdateFrame(0.0, 0, view, proj, eye)
    fi = f.frame_info()
    assert (fi.cube_lod, fi.ray_samples, fi.visibility_mask, fi.cube_size) == (1, ns, mask, S)
    f.Render(0, fx.Fluid.OPTIMIZED)
    f.Synchronize()
    cube_close(f.download(fx.FIELD_CUBEMAP), WIDE["cube_lod1_separate"])
    f.SetMaxSamples(48, nml)
    f.UpdateFrame(0.0, 0, view, proj, eye)
    f.Render(0, fx.Fluid.RAY_MARCH_CUBEMAP)
    f.Synchronize()
    cube_close(f.download(fx.FIELD_CUBEMAP), WIDE["cube_lod1_merged"])


def test_sh_transform_at_the_reference_size_vs_reference_binaries():
    """CSSHCubeMap + CSSHSum x 3 + CSSHNormalize at SH_TEX_SIZE = 256: the product computes the INTENDED reduction (every pass its own
    element count); the as-shipped one (LightProbeEZ.cpp:245-246) differs from it by the fixture's 1.4e-3 of the largest coefficient"""
    cube = np.repeat(np.repeat(WIDE["sh256_base32"], 8, axis=1), 8, axis=2)
    f = make((16, 16, 16))
    lp = fx.LightProbe(f)
    assert lp.Init(cube)
    lp.TransformSH()
    got = lp.GetSH()
    assert np.allclose(got, WIDE["sh256_intended"], rtol=2e-5, atol=2e-6)
    scale = np.abs(WIDE["sh256_intended"]).max()
    assert np.abs(got - WIDE["sh256_as_shipped"]).max() / scale > 1e-4
