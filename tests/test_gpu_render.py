"""GPU parity tests of the cube-map-space ray march and the SH light probe vs the CPU oracle.
Bars: light map = packed R11G11B10F -> decoded values equal the oracle's except where a 1-ulp
difference of rsq/division flips a rounding (rare; asserted as a fraction); cube map = RGBA8 ->
at most +-1 LSB (compare renders with an 8-bit tolerance); SH = 1e-5."""
import numpy as np
import pytest

import fluidx12_amd as fx
from fluidx12_amd import capi
from oracle import orc

pytestmark = pytest.mark.gpu
f32 = np.float32


def smoke_state(X, steps=10, seed=None):
    """a density field with structure: a few simulation steps of the oracle (+ optional noise blob)"""
    s = orc.Sim(X, X, X, iters=20)
    for _ in range(steps):
        s.step()
    col = s.color.copy()
    if seed is not None:
        rng = np.random.default_rng(seed)
        z, y, x = np.meshgrid(*(np.arange(X),) * 3, indexing="ij")
        blob = np.exp(-(((x - X * .55) ** 2 + (y - X * .5) ** 2 + (z - X * .45) ** 2) / (X * .22) ** 2)).astype(f32)
        col += (blob[..., None] * rng.random((X, X, X, 4)) * np.array([.3, .5, .8, .6])).astype(f32)
        col = np.clip(col, 0, 1).astype(f32)
    return col


def setup(X, col, vw=640, vh=480, storage="fp32", sh=None, max_samples=(192, 64)):
    f = fx.Fluid()
    assert f.Init(vw, vh, (X, X, X), storage=storage)
    f.SetMaxSamples(*max_samples)
    if sh is not None:
        f.SetSH(sh)
    view, proj, eye = fx.default_camera(vw, vh)
    f.upload(fx.FIELD_COLOR, col)
    f.UpdateFrame(0.0, 0, view, proj, eye)          # dt = 0: parity stays, colour[parity] is what we uploaded
    fr, lod, rs, mask, _ = orc.update_frame(view, proj, eye, vw, vh, X, max_samples[0])
    fi = f.frame_info()
    assert (fi.cube_lod, fi.ray_samples, fi.visibility_mask) == (lod, rs, mask)
    if sh is not None:
        for i, v in enumerate(np.asarray(sh, f32).reshape(27)):
            fr.sh[i] = v
    return f, fr, lod, rs, mask


def cube_close(gpu, ref, max_lsb=1, frac=0.02):
    d = np.abs(gpu.astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= max_lsb, int(d.max())
    assert (d > 0).mean() <= frac, float((d > 0).mean())


@pytest.mark.parametrize("X,vp", [(32, (640, 480)), (48, (1920, 1080)), (64, (200, 150))])
def test_update_frame_matches_oracle(X, vp):
    f, fr, lod, rs, mask = setup(X, np.zeros((X, X, X, 4), f32), *vp)
    assert f.frame_info().cube_size == X >> lod


@pytest.mark.parametrize("X", [32, 40])
def test_separate_light_pass(X):
    col = smoke_state(X, 8, seed=3)
    f, fr, lod, rs, mask = setup(X, col)
    f.Render(0, fx.Fluid.OPTIMIZED)
    f.Synchronize()
    lm_ref = orc.raymarch_light(col, fr, 64, False, 2)
    lm = f.download(fx.FIELD_LIGHTMAP)
    assert (lm != lm_ref).mean() < 1e-3                      # rare R11G11B10 rounding flips only
    assert np.abs(lm - lm_ref).max() <= np.abs(lm_ref).max() * 2.0 ** -5
    cf, cu = orc.raymarch_view(col, lm_ref, fr, X >> lod, mask, rs, 64, False, True)
    cube = f.download(fx.FIELD_CUBEMAP)
    assert cu[..., 3].max() > 50                             # the test volume is actually visible
    cube_close(cube, cu)
    # culled faces are never written (CSRayMarch.hlsl:102): they stay at their cleared value
    for face in range(6):
        if not (mask >> face) & 1:
            assert not cube[face].any()


def test_merged_march():
    X = 32
    col = smoke_state(X, 8, seed=4)
    f, fr, lod, rs, mask = setup(X, col, max_samples=(96, 24))
    f.Render(0, fx.Fluid.RAY_MARCH_CUBEMAP)
    f.Synchronize()
    _, cu = orc.raymarch_view(col, None, fr, X >> lod, mask, rs, 24, False, False)
    cube_close(f.download(fx.FIELD_CUBEMAP), cu)


def synthetic_radiance(N):
    """L(dir) = max(dir.y, 0) * (1, .9, .8) + 0.1"""
    cube = np.empty((6, N, N, 3), f32)
    idx = np.arange(N, dtype=f32)
    px, py = np.meshgrid(idx - N / 2 + 0.5, -(idx - N / 2 + 0.5))     # x varies along columns, y along rows
    pz = np.full_like(px, N / 2)
    dirs = [(pz, py, -px), (-pz, py, px), (px, pz, -py), (px, -pz, py), (px, py, pz), (-px, py, -pz)]
    for f_, (dx, dy, dz) in enumerate(dirs):
        n = dy / np.sqrt(dx * dx + dy * dy + dz * dz)
        cube[f_] = np.maximum(n, 0)[..., None] * np.array([1, .9, .8], f32) + f32(0.1)
    return cube


@pytest.mark.parametrize("N", [16, 64, 256])
def test_sh_transform(N):
    cube = synthetic_radiance(N)
    f = fx.Fluid()
    assert f.Init(640, 480, (16, 16, 16))
    lp = fx.LightProbe(f)
    assert lp.Init(cube)
    lp.TransformSH()
    sh = lp.GetSH()
    ref = orc.sh_transform(cube)
    assert np.allclose(sh, ref, rtol=1e-5, atol=1e-6)
    # the up direction is brighter than down for this sky
    assert orc.sh_irradiance(sh, [0, 1, 0])[0] > orc.sh_irradiance(sh, [0, -1, 0])[0]


def test_sh_constant_radiance_known_answer():
    N = 32
    c = np.array([0.3, 0.6, 0.9], f32)
    cube = np.broadcast_to(c, (6, N, N, 3)).copy()
    f = fx.Fluid()
    assert f.Init(640, 480, (16, 16, 16))
    lp = fx.LightProbe(f)
    lp.Init(cube)
    lp.TransformSH()
    sh = lp.GetSH()
    assert np.allclose(sh[0], c * 0.2820948 * 4 * np.pi, rtol=2e-5)
    assert np.abs(sh[1:]).max() < 2e-4


@pytest.mark.parametrize("flags", [fx.Fluid.OPTIMIZED, fx.Fluid.RAY_MARCH_CUBEMAP])
def test_gi_with_light_probe(flags):
    X = 32
    col = smoke_state(X, 8, seed=5)
    sh = orc.sh_transform(synthetic_radiance(32))
    f, fr, lod, rs, mask = setup(X, col, sh=sh, max_samples=(96, 24))
    f.Render(0, flags)
    f.Synchronize()
    if flags == fx.Fluid.OPTIMIZED:
        lm_ref = orc.raymarch_light(col, fr, 24, True, 2)
        lm = f.download(fx.FIELD_LIGHTMAP)
        assert (lm != lm_ref).mean() < 5e-3
        _, cu = orc.raymarch_view(col, lm_ref, fr, X >> lod, mask, rs, 24, True, True)
    else:
        _, cu = orc.raymarch_view(col, None, fr, X >> lod, mask, rs, 24, True, False)
    cube_close(f.download(fx.FIELD_CUBEMAP), cu)


def test_fp16_colour_render():
    X = 32
    col = smoke_state(X, 8, seed=6).astype(np.float16).astype(f32)
    f, fr, lod, rs, mask = setup(X, col, storage="fp16")
    f.Render(0, fx.Fluid.OPTIMIZED)
    f.Synchronize()
    lm_ref = orc.raymarch_light(col, fr, 64, False, 2)
    _, cu = orc.raymarch_view(col, lm_ref, fr, X >> lod, mask, rs, 64, False, True)
    cube_close(f.download(fx.FIELD_CUBEMAP), cu)


def test_render_errors():
    f = fx.Fluid()
    assert f.Init(640, 480, (16, 16, 16))
    with pytest.raises(fx.FluidxError):
        f.Render(0, fx.Fluid.OPTIMIZED)                  # UpdateFrame with a camera first
    f2 = fx.Fluid()
    assert f2.Init(640, 480, (32, 32, 1))
    with pytest.raises(fx.FluidxError):
        f2.Render(0, fx.Fluid.OPTIMIZED)                 # no frame yet


def test_config3_render_properties():
    """256^3, default camera at 1920x1080: LOD 0, 192 samples, faces {+X,-X,-Y,+Z};
    full-size checks are properties, not an oracle replay."""
    X = 256
    f = fx.Fluid()
    assert f.Init(1920, 1080, (X, X, X))
    view, proj, eye = fx.default_camera(1920, 1080)
    for k in range(24):
        f.UpdateFrame(f32(f.default_time_step()), k % 3, view, proj, eye)
        f.Simulate(k % 3)
    f.Render(0, fx.Fluid.OPTIMIZED)
    f.Synchronize()
    fi = f.frame_info()
    assert (fi.cube_lod, fi.ray_samples, fi.visibility_mask) == (0, 192, 0x1B)
    cube = f.download(fx.FIELD_CUBEMAP)
    assert cube.shape == (6, X, X, 4)
    assert not cube[2].any() and not cube[5].any()       # +Y and -Z are culled
    assert cube[..., 3].max() > 0                        # smoke is visible
    lm = f.download(fx.FIELD_LIGHTMAP)
    amb = np.float32(1.5 * np.pi)
    # empty voxels receive exactly light colour + ambient (shadow = 1): (1,.7,.3)*3pi + 1.5pi, R11G11B10-rounded
    corner = lm[0, 0, 0]
    assert np.allclose(corner, np.array([1, .7, .3]) * 3 * np.pi + amb, rtol=2 ** -5)
    # idempotence: rendering the same state twice gives the same cube map
    f.Render(0, fx.Fluid.OPTIMIZED)
    f.Synchronize()
    assert np.array_equal(cube, f.download(fx.FIELD_CUBEMAP))


# ---- cube map -> screen resolve (row f-1: Fluid::renderCube, PSRayCastCube.hlsl, PSCube.hlsli) ---------------------------
def oracle_frame_of(f, view, proj, eye, X):
    """the oracle's frame constants + the library's own WorldViewProjI, so that both run on identical inputs"""
    vw, vh = f.viewport
    fr = orc.update_frame(view, proj, eye, vw, vh, X, 192)[0]
    fi = f.frame_info()
    wvp_i = np.array(list(fi.world_view_proj_i), f32).reshape(4, 4)
    ref = orc.world_view_proj_inverse(view, proj)
    assert np.abs(wvp_i - ref).max() <= 1e-5 * np.abs(ref).max()       # two fp32 restatements of XMMatrixInverse
    return fr, wvp_i


@pytest.mark.parametrize("N,vp,seed", [(16, (160, 120), 1), (8, (96, 72), 2), (32, (320, 200), 3)])
def test_cube_resolve_equals_oracle(N, vp, seed):
    """k_resolve_cube on a fully random cube map (every face, seamless edge and corner carries data):
    SV_TARGET and the blended RGBA8 target equal the oracle bit for bit"""
    rng = np.random.default_rng(seed)
    cube = rng.integers(0, 256, (6, N, N, 4), dtype=np.uint8)
    f = fx.Fluid()
    assert f.Init(vp[0], vp[1], (N, N, N))
    view, proj, eye = fx.default_camera(*vp)
    f.UpdateFrame(0.0, 0, view, proj, eye)
    assert f.frame_info().cube_size == N
    f.upload(fx.FIELD_CUBEMAP, cube)
    assert np.array_equal(f.download(fx.FIELD_CUBEMAP), cube)
    fr, wvp_i = oracle_frame_of(f, view, proj, eye, N)
    out, cov = orc.resolve_cube(cube, fr, wvp_i, vp[0], vp[1])
    f.ClearRenderTarget((0.2, 0.2, 0.2, 0.0))
    f.RenderCube(0)
    f.Synchronize()
    got = f.download(fx.FIELD_TARGET_FLOAT)
    assert cov.mean() > 0.1
    assert np.array_equal(got.view(np.uint32), out.view(np.uint32))
    target = np.empty((vp[1], vp[0], 4), np.uint8)
    target[...] = (51, 51, 51, 0)
    assert np.array_equal(f.download(fx.FIELD_TARGET), orc.blend_premultiplied(out, cov, target))
    # the blend accumulates: a second resolve composites over the first (the caller clears once per frame)
    f.RenderCube(0)
    f.Synchronize()
    twice = orc.blend_premultiplied(out, cov, orc.blend_premultiplied(out, cov, target))
    assert np.array_equal(f.download(fx.FIELD_TARGET), twice)


def test_render_to_target_pipeline():
    """simulate -> Render(OPTIMIZED) -> renderCube at the demo's 640x480: the whole Fluid::Render of the cube path.
    The resolved image equals the oracle's resolve of the downloaded cube map; uncovered pixels keep the clear colour."""
    X, vp = 32, (640, 480)
    f, fr, lod, rs, mask = setup(X, smoke_state(X, 8, seed=5), *vp)
    view, proj, eye = fx.default_camera(*vp)
    f.ClearRenderTarget()
    f.Render(0, fx.Fluid.OPTIMIZED, to_target=True)
    f.Synchronize()
    cube = f.download(fx.FIELD_CUBEMAP)
    fr2, wvp_i = oracle_frame_of(f, view, proj, eye, X)
    out, cov = orc.resolve_cube(cube, fr2, wvp_i, *vp)
    img = f.download(fx.FIELD_TARGET)
    target = np.empty((vp[1], vp[0], 4), np.uint8)
    target[...] = (51, 51, 51, 0)
    assert np.array_equal(img, orc.blend_premultiplied(out, cov, target))
    assert 0.02 < cov.mean() < 0.9 and (img[~cov.astype(bool)] == (51, 51, 51, 0)).all()
    assert img[cov.astype(bool)][:, 3].max() > 50


def test_render_cube_errors():
    f = fx.Fluid()
    assert f.Init(64, 64, (16, 16, 16))
    with pytest.raises(fx.FluidxError):
        f.RenderCube(0)                                   # no view yet
    with pytest.raises(fx.FluidxError):
        f.download(fx.FIELD_TARGET)                       # no target yet
    f2 = fx.Fluid()
    assert f2.Init(64, 64, (16, 16, 1))
    f2.UpdateFrame(0.0, 0)
    with pytest.raises(fx.FluidxError):
        f2.RenderCube(0)                                  # 2D has no cube map


# ---- direct screen-space march (row f-2: Fluid::rayCastDirect / rayCastVDirect, PSRayCast.hlsl, PSRayCastV.hlsl) ----------
def rgba8_close(got, ref, frac=0.002):
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() <= frac, (int(d.max()), float((d > 0).mean()))


@pytest.mark.parametrize("flags,use_sh", [(fx.Fluid.SEPARATE_LIGHT_PASS, False), (fx.Fluid.RAY_MARCH_DIRECT, False),
                                          (fx.Fluid.RAY_MARCH_DIRECT, True)])
def test_direct_ray_cast_equals_oracle(flags, use_sh):
    """Render() without RAY_MARCH_CUBEMAP: one ray per screen pixel.  SV_TARGET equals the oracle (bit for bit up to the
    light map's rare R11G11B10 rounding flips in the separate mode) and the blended target equals the oracle's blend."""
    X, vp = 32, (200, 150)
    sh = (np.random.default_rng(4).random((9, 3)) * np.array([[2.0]] + [[0.5]] * 8)).astype(f32) if use_sh else None
    col = smoke_state(X, 8, seed=6)
    f, fr, lod, rs, mask = setup(X, col, *vp, sh=sh, max_samples=(48, 16))
    view, proj, eye = fx.default_camera(*vp)
    fr2, wvp_i = oracle_frame_of(f, view, proj, eye, X)
    separate = bool(flags & fx.Fluid.SEPARATE_LIGHT_PASS)
    if separate:
        lm = orc.raymarch_light(col, fr, 16, use_sh, 2)
        out, cov = orc.raycast_direct(col, lm, fr, wvp_i, vp[0], vp[1], rs, 16, use_sh, True)
    else:
        out, cov = orc.raycast_direct(col, None, fr, wvp_i, vp[0], vp[1], 48, 16, use_sh, False)
    f.ClearRenderTarget()
    f.Render(0, flags)
    f.Synchronize()
    got = f.download(fx.FIELD_TARGET_FLOAT)
    assert 0.05 < cov.mean() < 0.9 and out[..., 3].max() > 0.3
    if separate:
        assert np.mean(got != out) < 2e-3 and np.abs(got - out).max() < 0.05
    else:
        assert np.array_equal(got.view(np.uint32), out.view(np.uint32))
    target = np.empty((vp[1], vp[0], 4), np.uint8)
    target[...] = (51, 51, 51, 0)
    rgba8_close(f.download(fx.FIELD_TARGET), orc.blend_premultiplied(out, cov, target))


def test_direct_and_cube_paths_show_the_same_picture():
    """the paper's comparison: the cube-map-space path (march 4 x N^2 texels, then resolve) approximates the direct
    march of every pixel -- same silhouette, close colours"""
    X, vp = 48, (320, 240)
    f, fr, lod, rs, mask = setup(X, smoke_state(X, 10, seed=7), *vp)
    f.ClearRenderTarget()
    f.Render(0, fx.Fluid.SEPARATE_LIGHT_PASS)
    f.Synchronize()
    direct = f.download(fx.FIELD_TARGET).astype(np.int32)
    f.ClearRenderTarget()
    f.Render(0, fx.Fluid.OPTIMIZED, to_target=True)
    f.Synchronize()
    cube = f.download(fx.FIELD_TARGET).astype(np.int32)
    covered = (direct[..., 3] > 8) | (cube[..., 3] > 8)
    assert covered.mean() > 0.05
    assert np.abs(direct - cube)[covered].mean() < 6.0          # of 255


@pytest.mark.parametrize("storage", ["fp32", "fp16"])
def test_2d_visualiser_equals_oracle(storage):
    """Render() of a 2-D grid = Fluid::visualizeColor (PSVisualizeColor.hlsl): tone-mapped colour[parity] on the target"""
    X, vp = 64, (200, 120)
    f = fx.Fluid()
    assert f.Init(vp[0], vp[1], (X, X, 1), storage=storage, jacobi_iters=20)
    for k in range(12):
        f.UpdateFrame(f32(f.default_time_step()), k % 3)
        f.Simulate(k % 3)
    f.ClearRenderTarget()
    f.Render(0, fx.Fluid.OPTIMIZED)
    f.Synchronize()
    col = f.download(fx.FIELD_COLOR)
    assert col[..., 3].max() > 0.2
    out = orc.visualize_color(col, *vp)
    got = f.download(fx.FIELD_TARGET_FLOAT)
    assert np.array_equal(got.view(np.uint32), out.view(np.uint32))
    target = np.empty((vp[1], vp[0], 4), np.uint8)
    target[...] = (51, 51, 51, 0)
    assert np.array_equal(f.download(fx.FIELD_TARGET), orc.blend_premultiplied(out, np.ones(vp[::-1], np.uint8), target))


@pytest.mark.parametrize("storage,use_sh", [("fp32", False), ("fp16", True)])
def test_empty_space_skipping_changes_no_bit(storage, use_sh):
    """the accelerated marches (FX_OPT_RENDER_ACCEL, the default: occupancy masks in the LDS, alpha side volume, compacted light
    voxels) only skip gathers whose result is known (all taps 0, or all taps <= the 0.01 threshold of the view march): light
    map, both cube-map marches and the direct marches are bit-identical to the plain kernels, where every sample gathers"""
    X, vp = 48, (240, 180)
    col = smoke_state(X, 10, seed=11)
    sh = (np.random.default_rng(5).random((9, 3)) * np.array([[2.0]] + [[0.5]] * 8)).astype(f32) if use_sh else None

    def run(on):
        f, fr, lod, rs, mask = setup(X, col, *vp, storage=storage, sh=sh, max_samples=(64, 24))
        f.set_option(capi.OPT_RENDER_ACCEL, 1 if on else 0)
        out = []
        for flags in (fx.Fluid.OPTIMIZED, fx.Fluid.RAY_MARCH_CUBEMAP):
            f.Render(0, flags)
            f.Synchronize()
            out += [f.download(fx.FIELD_LIGHTMAP), f.download(fx.FIELD_CUBEMAP)]
        for flags in (fx.Fluid.SEPARATE_LIGHT_PASS, fx.Fluid.RAY_MARCH_DIRECT):
            f.ClearRenderTarget()
            f.Render(0, flags)
            f.Synchronize()
            out.append(f.download(fx.FIELD_TARGET_FLOAT))
        return out

    a, b = run(False), run(True)
    assert a[1][..., 3].max() > 30 and a[4][..., 3].max() > 0.2
    for u, v in zip(a, b):
        assert np.array_equal(u.view(np.uint8), v.view(np.uint8))


def test_environment_pass_equals_oracle():
    """the light probe's sky pass (PSEnvironment) onto the render target: bit-identical to the oracle given the library's own
    ScreenToWorld; then the volume composites over it exactly like over a cleared target"""
    X, vp = 32, (200, 150)
    rng = np.random.default_rng(9)
    cube = (rng.random((6, 16, 16, 3)) ** 3 * 4.0).astype(f32)
    f, fr, lod, rs, mask = setup(X, smoke_state(X, 8, seed=4), *vp)
    view, proj, eye = fx.default_camera(*vp)
    with pytest.raises(fx.FluidxError):
        f.RenderEnvironment(0)                              # no cube yet
    f.SetEnvironment(cube)
    f.RenderEnvironment(0)
    f.Synchronize()
    fi = f.frame_info()
    s2w = np.array(list(fi.screen_to_world), f32).reshape(4, 4)
    want = orc.environment(cube, eye, s2w, *vp)
    got = f.download(fx.FIELD_TARGET_FLOAT)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    sky8 = f.download(fx.FIELD_TARGET)
    q = np.clip(want[..., :3], 0, 1)
    assert np.array_equal(sky8[..., :3], np.floor(q * f32(255) + f32(0.5)).astype(np.uint8)) and not sky8[..., 3].any()
    # the volume over the sky = the oracle's blend over the sky image
    f.Render(0, fx.Fluid.OPTIMIZED, to_target=True)
    f.Synchronize()
    cube_map = f.download(fx.FIELD_CUBEMAP)
    fr2, wvp_i = oracle_frame_of(f, view, proj, eye, X)
    out, cov = orc.resolve_cube(cube_map, fr2, wvp_i, *vp)
    assert np.array_equal(f.download(fx.FIELD_TARGET), orc.blend_premultiplied(out, cov, sky8))


# ---- full-size render parity (BASELINE configs 3 and 5) -------------------------------------------------
def developed_plume(X, steps, storage, vp, sh=None, camera=None):
    """`steps` frames of the HIP simulation from rest (the renderers' input is whatever the solver produced), then one paused
    frame with the camera: returns the context, its colour field and the oracle's frame constants"""
    f = fx.Fluid()
    assert f.Init(vp[0], vp[1], (X, X, X), storage=storage)
    view, proj, eye = camera if camera is not None else fx.default_camera(*vp)
    for k in range(steps):
        f.UpdateFrame(f32(f.default_time_step()), k % 3, view, proj, eye)
        f.Simulate(k % 3)
        if k == steps - 2:
            f.Render(k % 3, fx.Fluid.OPTIMIZED)       # a context that renders its frames: the last step's advection writes the render's side volume (the product's default flow)
    if sh is not None:
        f.SetSH(sh)
    f.UpdateFrame(0.0, 0, view, proj, eye)
    f.Synchronize()
    col = f.download(fx.FIELD_COLOR)
    fr, lod, rs, mask, _ = orc.update_frame(view, proj, eye, vp[0], vp[1], X, 192)
    fi = f.frame_info()
    assert (fi.cube_lod, fi.ray_samples, fi.visibility_mask) == (lod, rs, mask)
    if sh is not None:
        for i, v in enumerate(np.asarray(sh, f32).reshape(27)):
            fr.sh[i] = v
    return f, col, fr, lod, rs, mask


def check_render_against_oracle(f, col, fr, X, lod, rs, mask, use_sh):
    """OPTIMIZED (CSRayMarchL + CSRayMarchV) and the merged CSRayMarch of one state against the oracle's replay of the same state"""
    f.Render(0, fx.Fluid.OPTIMIZED)
    f.Synchronize()
    lm_ref = orc.raymarch_light(col, fr, 64, use_sh, 2)
    lm = f.download(fx.FIELD_LIGHTMAP)
    lit = col[..., 3] >= f32(0.01)
    assert 0.002 < lit.mean() < 0.5                                   # a plume, neither a puff nor fog
    assert (lm != lm_ref).mean() < 1e-3                               # rare R11G11B10 rounding flips only
    assert np.abs(lm - lm_ref).max() <= np.abs(lm_ref).max() * 2.0 ** -5
    _, cu = orc.raymarch_view(col, lm_ref, fr, X >> lod, mask, rs, 64, use_sh, True)
    cube = f.download(fx.FIELD_CUBEMAP)
    assert cu[..., 3].max() > 100
    cube_close(cube, cu, max_lsb=1, frac=0.01)
    f.Render(0, fx.Fluid.RAY_MARCH_CUBEMAP)
    f.Synchronize()
    _, cu2 = orc.raymarch_view(col, None, fr, X >> lod, mask, rs, 64, use_sh, False)
    cube_close(f.download(fx.FIELD_CUBEMAP), cu2, max_lsb=1, frac=0.01)
    for face in range(6):
        if not (mask >> face) & 1:
            assert not cube[face].any()


def test_config3_render_against_oracle():
    """BASELINE configs[2] at full size: 256^3 fp32 after 64 simulated frames, default camera at 1920x1080 (LOD 0, 4 x 256^2 rays,
    192 + 64 samples): light volume and cube map of both cube-map marches against the oracle -- light map equal up to rare
    R11G11B10 rounding flips, cube map <= 1 LSB on < 1 % of the texels"""
    X = 256
    f, col, fr, lod, rs, mask = developed_plume(X, 64, "fp32", (1920, 1080))
    assert (lod, rs, mask) == (0, 192, 0x1B)
    check_render_against_oracle(f, col, fr, X, lod, rs, mask, False)


def test_config5_render_against_oracle():
    """BASELINE configs[4] at full size: 256^3 with RGBA16F fields and the SH light probe (hasSH = 1: GI + AO rays)"""
    X = 256
    sh = orc.sh_transform(synthetic_radiance(64))
    f, col, fr, lod, rs, mask = developed_plume(X, 64, "fp16", (1920, 1080), sh=sh)
    check_render_against_oracle(f, col, fr, X, lod, rs, mask, True)


def test_render_at_coarser_cube_mip_against_oracle():
    """the reference's default grid (128^3) seen from further away: UpdateFrame picks cube-map LOD >= 1, the view march runs
    on a smaller mip with fewer samples; eye off the default axis so that another set of faces is visible"""
    X, vp = 128, (640, 480)
    eye = np.array([-50.0, 40.0, -100.0], f32)
    cam = (fx.look_at_lh(eye, [0, 0, 0], [0, 1, 0]), fx.perspective_fov_lh(f32(np.pi) / f32(4.0), vp[0] / float(vp[1]), 1.0, 1000.0), eye)
    f, col, fr, lod, rs, mask = developed_plume(X, 60, "fp32", vp, camera=cam)
    assert lod >= 1 and mask != 0x1B
    check_render_against_oracle(f, col, fr, X, lod, rs, mask, False)


@pytest.mark.parametrize("dims,storage,lds", [((64, 64, 64), "fp32", "2"), ((64, 64, 64), "fp16", "2"), ((72, 72, 40), "fp32", "2"), ((128, 128, 32), "fp16", "2"),
                                              ((64, 64, 64), "fp32", "0"), ((32, 32, 64), "fp16", "0"), ((72, 72, 40), "fp16", "0"), ((50, 50, 30), "fp32", "0")])
def test_alpha_side_volume_written_by_the_advection_changes_no_bit(dims, storage, lds, knob):
    """the advection writes the stored alpha, as fp32, into the render's side volume (k_advect_lds / k_advect_far <ALPHA>, k_advect_fast, k_advect) and the
    render's build pass then folds the block maxima from those 4 bytes per voxel instead of reading the texels: same light map, cube
    map and direct picture as with the side volume extracted by the build pass (ADVECT_ALPHA=0) and as the plain kernels; an upload
    over the colour field takes the mirror away"""
    knob("ADVECT_LDS", lds)                                           # "2": the staged path below the size where it pays; "0": the gather kernels (k_advect_fast / k_advect)
    vp = (320, 240)
    view, proj, eye = fx.default_camera(*vp)

    def run(alpha, accel=1):
        knob("ADVECT_ALPHA", "1" if alpha else "0")
        f = fx.Fluid()
        assert f.Init(vp[0], vp[1], dims, storage=storage)
        f.SetMaxSamples(96, 32)
        f.set_option(capi.OPT_RENDER_ACCEL, accel)
        out = []
        for k in range(14):
            f.UpdateFrame(f32(f.default_time_step()), k % 3, view, proj, eye)
            f.Simulate(k % 3)
            if k in (6, 7, 12, 13):                                   # the step behind a rendered frame is the one that writes the side volume
                f.Render(k % 3, fx.Fluid.OPTIMIZED)
                f.Synchronize()
                if k in (7, 13):
                    out += [f.download(fx.FIELD_LIGHTMAP), f.download(fx.FIELD_CUBEMAP)]
        f.ClearRenderTarget()
        f.Render(0, fx.Fluid.RAY_MARCH_DIRECT)
        f.Synchronize()
        out.append(f.download(fx.FIELD_TARGET_FLOAT))
        # another colour field over the current one: the side volume must be rebuilt from it
        col = f.download(fx.FIELD_COLOR)
        col2 = np.ascontiguousarray(col[::-1, :, ::-1])
        f.upload(fx.FIELD_COLOR, col2)
        f.Render(0, fx.Fluid.OPTIMIZED)
        f.Synchronize()
        out += [f.download(fx.FIELD_LIGHTMAP), f.download(fx.FIELD_CUBEMAP), f.download(fx.FIELD_VELOCITY), col]
        return out

    a, b, c = run(True), run(False), run(True, accel=0)
    assert a[1][..., 3].max() > 20 and a[3][..., 3].max() > 20
    for u, v, w in zip(a, b, c):
        assert np.array_equal(u.view(np.uint8), v.view(np.uint8))
        assert np.array_equal(u.view(np.uint8), w.view(np.uint8))


def test_light_fill_paths_alternate_without_a_trace(knob):
    """the build pass that also fills the light map (k_build_fill) appends to this render's list counters, which the PREVIOUS render's
    build pass cleared: alternate it with the three-pass path, the plain kernels, a light probe switched on and off and an upload, frame
    by frame -- every light map and cube map equals the plain kernels' of the same frame"""
    X, vp = 64, (320, 240)
    view, proj, eye = fx.default_camera(*vp)
    sh = (np.random.default_rng(3).random((9, 3)) * np.array([[2.0]] + [[0.5]] * 8)).astype(f32)

    def run(script):
        f = fx.Fluid()
        assert f.Init(vp[0], vp[1], (X, X, X), storage="fp32")
        f.SetMaxSamples(96, 32)
        out = []
        for k, (accel, fill, use_sh, upload) in enumerate(script):
            knob("LIGHT_FILL", "1" if fill else "0")
            f.set_option(capi.OPT_RENDER_ACCEL, 1 if accel else 0)
            f.SetSH(sh if use_sh else None)
            f.UpdateFrame(f32(f.default_time_step()), k % 3, view, proj, eye)
            f.Simulate(k % 3)
            if upload:
                col = f.download(fx.FIELD_COLOR)
                f.upload(fx.FIELD_COLOR, np.ascontiguousarray(col[:, ::-1]))
            f.Render(k % 3, fx.Fluid.OPTIMIZED)
            f.Synchronize()
            out += [f.download(fx.FIELD_LIGHTMAP), f.download(fx.FIELD_CUBEMAP)]
        return out

    #          accel fill   sh     upload
    script = [(1, 1, False, 0), (1, 1, False, 0), (1, 0, False, 0), (1, 1, False, 0), (0, 1, False, 0), (1, 1, False, 0), (1, 1, True, 0), (1, 1, True, 0),
              (1, 0, True, 0), (1, 1, True, 1), (1, 1, True, 0), (1, 1, False, 0), (1, 1, False, 0), (1, 1, False, 0)] + [(1, 1, False, 0)] * 6
    a = run(script)
    b = run([(0, 0, s_[2], s_[3]) for s_ in script])
    assert a[-1][..., 3].max() > 20
    for k, (u, v) in enumerate(zip(a, b)):
        assert np.array_equal(u.view(np.uint8), v.view(np.uint8)), k


@pytest.mark.parametrize("X,storage,use_sh", [(288, "fp32", False), (512, "fp16", True)])
def test_grids_whose_cell_mask_exceeds_the_lds_change_no_bit(X, storage, use_sh):
    """above 256^3 the 4^3-cell masks no longer fit the LDS budget: the marches hold a coarser level there and confirm a set bit against the
    fine occupancy grid (AccelVol<.., COARSE>); 288^3 takes the three-pass light volume, 512^3 the filling build pass.  Light map, cube
    map and the direct picture equal the plain kernels' bit for bit"""
    vp = (640, 360)
    sh = (np.random.default_rng(9).random((9, 3)) * np.array([[2.0]] + [[0.5]] * 8)).astype(f32) if use_sh else None
    f, col, fr, lod, rs, mask = developed_plume(X, 40, storage, vp, sh=sh)
    out = []
    for accel in (1, 0):
        f.set_option(capi.OPT_RENDER_ACCEL, accel)
        f.Render(0, fx.Fluid.OPTIMIZED)
        f.Synchronize()
        res = [f.download(fx.FIELD_LIGHTMAP), f.download(fx.FIELD_CUBEMAP)]
        f.ClearRenderTarget()
        f.Render(0, fx.Fluid.SEPARATE_LIGHT_PASS)
        f.Synchronize()
        res.append(f.download(fx.FIELD_TARGET_FLOAT))
        out.append(res)
    assert out[0][1][..., 3].max() > 20
    for u, v in zip(*out):
        assert np.array_equal(u.view(np.uint8), v.view(np.uint8))
