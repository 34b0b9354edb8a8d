"""CPU tests of the oracle itself: the invariants that follow from the reference's shader source
(SURVEY.md section 4), the cross-check against the independent numpy restatement (np_ref.py), and the
golden fixtures under tests/golden/.  No GPU, no product code."""
import numpy as np
import pytest

from oracle import orc
import np_ref

f32 = np.float32


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = np.sqrt(((a - b) ** 2).sum())
    n = np.sqrt((b ** 2).sum())
    return d / n if n > 0 else d


def rand_state(X, Y, Z, seed=0, scale=0.5):
    rng = np.random.default_rng(seed)
    vel = (rng.standard_normal((3, Z, Y, X)) * scale).astype(f32)
    col = rng.random((Z, Y, X, 4)).astype(f32)
    p = rng.standard_normal((Z, Y, X)).astype(f32)
    return vel, col, p


# ---- storage formats --------------------------------------------------------------------------------
def test_half_conversion_matches_ieee():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.standard_normal(50000).astype(f32) * f32(s) for s in (1e-8, 1e-5, 1e-3, 1, 100, 7e4)])
    x = np.concatenate([x, np.array([0, -0.0, 65504, 65519.9, 65520, 2 ** -24, 2 ** -25, 2 ** -25 * 1.0001, 6.1e-5], f32)])
    out = np.empty_like(x)
    orc.lib().orc_quantize_half(orc._fp(x), orc._fp(out), x.size)
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16).astype(f32)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))


def test_r11g11b10_roundtrip():
    L = orc.lib()
    out = np.empty(3, f32)
    for v in (0.0, 1.0, 0.5, 3.0, 14.1371, 9.42, 4.712389, 1e-3, 6.1e-5, 1e-6, 65024.0):
        u = L.orc_pack_r11g11b10(v, v, v)
        L.orc_unpack_r11g11b10(u, orc._fp(out))
        assert abs(out[0] - v) <= v * 2.0 ** -7 + 2.0 ** -21     # 6-bit mantissa, RNE
        assert abs(out[2] - v) <= v * 2.0 ** -6 + 2.0 ** -20     # 5-bit mantissa
        u2 = L.orc_pack_r11g11b10(out[0], out[1], out[2])         # idempotent
        assert u2 == u
    assert L.orc_pack_r11g11b10(-1.0, -0.0, 0.0) == 0              # negatives clamp to 0


# ---- invariants ------------------------------------------------------------------
def test_dt_zero_is_identity():
    vel, col, p = rand_state(16, 16, 16, 2)
    vo, co = orc.advect(vel, col, 0.0)
    assert np.array_equal(vo, vel) and np.array_equal(co, col)


def test_zero_velocity_outside_ball_only_dissipates():
    X = 32
    vel = np.zeros((3, X, X, X), f32)
    col = np.random.default_rng(3).random((X, X, X, 4)).astype(f32)
    dt = f32(2.0 / X)
    vo, co = orc.advect(vel, col, dt)
    z, y, x = np.meshgrid(*(np.arange(X),) * 3, indexing="ij")
    d = np.sqrt(((x + .5) / X - .5) ** 2 + ((y + .5) / X - .1) ** 2 + ((z + .5) / X - .5) ** 2)
    outside = d > 1.0 / 16 + 1e-6
    atten = max(f32(1) - f32(0.2) * dt, f32(0))
    assert np.array_equal(co[outside], (col * atten)[outside])
    assert np.all(vo[:, outside] == 0)
    inside = d < 1.0 / 16 - 1e-6
    assert np.all(vo[1][inside] > 0)                 # buoyancy acts inside the impulse ball only


def test_linear_field_divergence():
    X = 16
    a, b, c = 0.25, -0.5, 0.75
    z, y, x = np.meshgrid(*(np.arange(X, dtype=f32),) * 3, indexing="ij")
    vel = np.stack([a * x, b * y, c * z]).astype(f32)
    d = orc.divergence(vel)
    assert np.allclose(d[1:-1, 1:-1, 1:-1], a + b + c, atol=1e-6)
    assert np.allclose(d[4, 4, 0], 0.5 * a + b + c, atol=1e-6)       # clamped neighbour halves that term


def test_constant_pressure_is_fixed_point():
    X = 12
    p = np.full((X, X, X), 3.25, f32)
    b = np.zeros_like(p)
    q, _ = orc.jacobi(p, b, 5)
    assert np.allclose(q, 3.25, rtol=0, atol=1e-6)
    vel = np.random.default_rng(4).standard_normal((3, X, X, X)).astype(f32) * f32(1e-3)
    out = orc.project(vel, p)
    # gradient of a constant is 0: only the wall factor may change u
    inner = (slice(None), slice(2, -2), slice(2, -2), slice(2, -2))
    assert np.array_equal(out[inner], vel[inner])


def test_rotational_symmetry_after_one_step():
    s = orc.Sim(32, 32, 32, iters=40)
    s.step()
    ux, uy, uz = s.velocity
    scale = np.abs(s.velocity).max()
    assert np.abs(ux + ux[::-1, :, ::-1]).max() < 1e-5 * scale
    assert np.abs(uz + uz[::-1, :, ::-1]).max() < 1e-5 * scale
    assert np.abs(uy - uy[::-1, :, ::-1]).max() < 1e-5 * scale


def test_wall_factor():
    X = 64
    vel = np.zeros((3, X, X, X), f32)
    vel[0] = 1.0                                    # moving towards +x everywhere
    out = orc.project(vel, np.zeros((X, X, X), f32))
    pos = (np.arange(X) + 0.5) / X * 2 - 1
    expect = np.where(pos > 0, np.clip((0.97 - np.abs(pos)) / 0.03, -1, 1), 1.0)
    assert np.allclose(out[0][5, 7, :], expect, atol=2e-5)


def test_faithful_mode_freezes():
    s = orc.Sim(24, 24, 24, iters=64, mode=1)
    s.step()
    b = s.b.copy()
    p0 = np.zeros_like(s.p)
    _, sweeps = orc.jacobi(p0, b, 64, mode=1)
    assert 1 <= sweeps <= 64


# ---- cross-check against the independent numpy restatement -----------------------------------------------
@pytest.mark.parametrize("dims,mirror", [((24, 24, 24), False), ((20, 20, 12), True), ((32, 32, 1), False)])
def test_oracle_matches_numpy_restatement(dims, mirror):
    X, Y, Z = dims
    vel, col, p = rand_state(X, Y, Z, 5, scale=0.8)
    dt = f32((2.0 if Z > 1 else 1.0) / Y)
    vo, co = orc.advect(vel, col, dt, address=int(mirror))
    vn, cn = np_ref.advect(vel, col, dt, mirror)
    assert rel_l2(vo, vn) < 2e-6 and rel_l2(co, cn) < 2e-6
    b = orc.divergence(vo)
    assert rel_l2(b, np_ref.divergence(vo)) < 1e-6
    q, _ = orc.jacobi(p, b, 20)
    assert rel_l2(q, np_ref.jacobi(p, b, 20)) < 2e-6
    assert rel_l2(orc.project(vo, q), np_ref.project(vo, q)) < 2e-6


def test_rollout_matches_numpy_restatement():
    X = 24
    s = orc.Sim(X, X, X, iters=20)
    vel = np.zeros((3, X, X, X), f32); col = np.zeros((X, X, X, 4), f32); p = np.zeros((X, X, X), f32)
    for _ in range(3):
        s.step()
        vel, col, p = np_ref.step(vel, col, p, s.default_dt(), 20)
    assert rel_l2(s.velocity, vel) < 1e-4
    assert rel_l2(s.color, col) < 1e-4


# ---- host rules ---------------------------------------------------------------------------------------------
def test_update_frame_default_camera():
    v, p, e = orc.default_camera(1920, 1080)
    fr, lod, rs, mask, edge = orc.update_frame(v, p, e, 1920, 1080, 256)
    assert (lod, rs, mask) == (0, 192, 0x1B)        # faces {+X,-X,-Y,+Z}
    assert abs(edge - 853.9) < 0.5
    assert orc.update_frame(v, p, e, 1920, 1080, 512)[1] == 1
    v, p, e = orc.default_camera(800, 800)
    assert abs(orc.update_frame(v, p, e, 800, 800, 128)[4] - 632.5) < 0.5
    assert np.allclose(list(fr.world_i), [0.1, 0, 0, 0, 0, 0.1, 0, 0, 0, 0, 0.1, 0])


# ---- spherical harmonics -----------------------------------------------------------------------------------
def test_sh_of_constant_radiance():
    N = 32
    c = np.array([0.3, 0.6, 0.9], f32)
    cube = np.broadcast_to(c, (6, N, N, 3)).copy()
    sh = orc.sh_transform(cube)
    assert np.allclose(sh[0], c * 0.2820948 * 4 * np.pi, rtol=2e-5)
    assert np.abs(sh[1:]).max() < 2e-4
    for n in ([0, 1, 0], [0.6, 0, 0.8], [-1, 0, 0]):
        assert np.allclose(orc.sh_irradiance(sh, n), np.pi * c, rtol=2e-4)


def test_sh_quirk_differs_slightly():
    N = 256                                         # the reference's hard-wired face size
    rng = np.random.default_rng(7)
    cube = rng.random((6, N, N, 3)).astype(f32)
    a, b = orc.sh_transform(cube), orc.sh_transform(cube, quirk=True)
    assert not np.array_equal(a, b)                 # 20 stale partials re-added in pass 3
    assert rel_l2(a, b) < 1e-2
