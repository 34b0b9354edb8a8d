"""GPU parity tests of the reference's OWN pressure solve (FX_JACOBI_FAITHFUL: CSPoisson.hlsli:8-26 under CSProject3D.hlsl:13, at most 64
sweeps, per-cell early-out at |dx| < 1e-3) on the sparse solver of fx_jacobi_freeze.hip: a dense first sweep + launches of <= 4
levels over the 32 x 8 x 8 tiles that still relax.  Bar: BIT-EXACT against the oracle's lock-step replay (orc_jacobi mode 1), the same
number of executed sweeps, and bit-identical to the one-sweep-per-launch kernel with the byte mask (k_jacobi_generic)."""
import os

import numpy as np
import pytest

import fluidx12_amd as fx
from oracle import orc

pytestmark = pytest.mark.gpu
f32 = np.float32


def make(dims, **kw):
    f = fx.Fluid()
    assert f.Init(0, 0, dims, **kw), f.last_status
    return f


def plume_like(X, Y, Z, seed, amp=0.05):
    """pressure and divergence whose sweeps move by more than 1e-3 only in a blob (plus a few specks), like the smoke solver's:
    most cells freeze in the first sweep, the rest over tens of sweeps"""
    rng = np.random.default_rng(seed)
    z, y, x = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
    c = rng.uniform(0.3, 0.7, 3)
    r2 = ((x + .5) / X - c[0]) ** 2 + ((y + .5) / Y - c[1]) ** 2 + ((z + .5) / Z - c[2]) ** 2
    w = np.exp(-r2 / 0.02)
    p = (rng.standard_normal((Z, Y, X)) * (amp * w + 2e-4)).astype(f32)
    b = (rng.standard_normal((Z, Y, X)) * (amp * w + 1e-4)).astype(f32)
    for _ in range(6):                                   # specks far from the blob: tiles that wake up alone
        zz, yy, xx = rng.integers(0, Z), rng.integers(0, Y), rng.integers(0, X)
        b[zz, yy, xx] += f32(0.3)
    return p, b


@pytest.fixture
def knobs(knob):
    """dict-style access to the launcher switches (tests/conftest.py `knob`): knobs["FLUIDX_FREEZE_T"] = "2" """
    class Setter:
        def __setitem__(self, key, value):
            knob(key[len("FLUIDX_"):] if key.startswith("FLUIDX_") else key, value)
    return Setter()


def solve(dims, p, b, iters):
    f = make(dims, jacobi_iters=iters, jacobi_mode="faithful")
    f.upload(fx.FIELD_PRESSURE, p)
    f.upload(fx.FIELD_DIVERGENCE, b)
    f.timing_read(True)
    f.Jacobi(iters)
    f.Synchronize()
    t = f.timing_read(True)
    out = f.download(fx.FIELD_PRESSURE)
    f.Release()
    return out, t


DIMS = [(32, 32, 32), (64, 64, 40), (40, 40, 12), (8, 8, 8), (36, 36, 9), (96, 96, 17), (4, 4, 3), (160, 160, 24),
        (150, 150, 20), (30, 30, 12), (37, 37, 9), (6, 6, 5), (67, 67, 11), (5, 5, 8)]        # rows that are no multiple of four cells: a short last quad


@pytest.mark.parametrize("dims", DIMS)
@pytest.mark.parametrize("iters", [1, 2, 5, 6, 17, 64])
def test_freeze_solver_equals_oracle(dims, iters, knobs):
    X, Y, Z = dims
    p, b = plume_like(X, Y, Z, 100 + X + iters)
    want, k = orc.jacobi(p, b, iters, mode=1)
    got, t = solve(dims, p, b, iters)
    assert np.array_equal(got, want)
    assert (t.freeze_solves, t.freeze_sweeps) == (1, k)


@pytest.mark.parametrize("T,NT,WGS", [(1, 512, 1024), (2, 256, 7), (3, 512, 64), (4, 256, 1), (3, 256, 100000), (4, 1024, 512), (2, 1024, 16)])
def test_freeze_solver_every_launch_shape(T, NT, WGS, knobs):
    """levels per launch, threads per workgroup and workgroups per launch change nothing"""
    knobs["FLUIDX_FREEZE_T"], knobs["FLUIDX_FREEZE_NT"], knobs["FLUIDX_FREEZE_WGS"] = str(T), str(NT), str(WGS)
    for dims, iters in (((64, 64, 24), 64), ((40, 40, 20), 23), ((32, 32, 9), 7)):
        p, b = plume_like(*dims, seed=7 * T + iters)
        want, k = orc.jacobi(p, b, iters, mode=1)
        got, t = solve(dims, p, b, iters)
        assert np.array_equal(got, want), (dims, iters)
        assert t.freeze_sweeps == k


@pytest.mark.parametrize("dims", [(128, 128, 128), (256, 256, 64), (150, 150, 150)])
def test_freeze_solver_at_reference_sizes(dims):
    """the reference's default grid (FluidX12.cpp:44), a 256-wide slab of the headline grid and its GI preset (Bin/FluidGI.bat:1: rows of
    150 cells), 64-sweep cap"""
    X, Y, Z = dims
    p, b = plume_like(X, Y, Z, 5, amp=0.08)
    want, k = orc.jacobi(p, b, 64, mode=1)
    got, t = solve(dims, p, b, 64)
    assert np.array_equal(got, want)
    assert t.freeze_sweeps == k and 1 < k <= 64


def test_freeze_solver_all_frozen_and_never_frozen():
    dims = (32, 32, 16)
    z = np.zeros(dims[::-1], f32)
    got, t = solve(dims, z, z, 64)                        # nothing moves: one sweep, like the oracle
    assert np.array_equal(got, z) and t.freeze_sweeps == 1
    rng = np.random.default_rng(3)
    p = rng.standard_normal(dims[::-1]).astype(f32) * 50
    b = rng.standard_normal(dims[::-1]).astype(f32) * 50
    want, k = orc.jacobi(p, b, 9, mode=1)                  # far from converged: the cap decides
    got, t = solve(dims, p, b, 9)
    assert k == 9 and t.freeze_sweeps == 9 and np.array_equal(got, want)


@pytest.mark.parametrize("storage,address,dims", [("fp16", "clamp", (64, 64, 64)), ("fp32", "mirror", (64, 64, 64)), ("fp16", "clamp", (50, 50, 50))])
def test_freeze_fast_path_equals_generic_kernel_over_steps(storage, address, dims, knobs):
    """whole steps in the reference's configuration (64-cap, early-out, RGBA16F / fp32): the sparse solver against the one-sweep
    kernel with the byte mask -- every field bit-identical after 12 steps, pressure buffer rotation included"""
    kw = dict(storage=storage, jacobi_iters=64, jacobi_mode="faithful", advect_address=address)
    fast = make(dims, **kw)
    knobs["FLUIDX_FREEZE_FAST"] = "0"
    slow = make(dims, **kw)
    knobs["FLUIDX_FREEZE_FAST"] = "1"
    for k in range(12):
        for f in (fast, slow):
            f.UpdateFrame(f32(f.default_time_step()), k % 3)
            f.Simulate(k % 3)
        if k in (0, 5, 11):
            for fld in (fx.FIELD_PRESSURE, fx.FIELD_VELOCITY, fx.FIELD_COLOR):
                assert np.array_equal(fast.download(fld), slow.download(fld)), (k, fld)
    t = fast.timing_read(True)
    assert t.freeze_solves == 12 and 12 < t.freeze_sweeps <= 12 * 64


def test_freeze_rollout_matches_oracle_128():
    """one step = advect + divergence + faithful solve + project at 128^3, RGBA16F storage, against the oracle stage by stage"""
    X = 128
    s = orc.Sim(X, X, X, iters=64, mode=1, half=True)
    f = make((X, X, X), storage="fp16", jacobi_iters=64, jacobi_mode="faithful")
    for k in range(3):
        s.step()
        f.UpdateFrame(f32(f.default_time_step()), 0)
        f.Simulate(0)
    # step 4 stage by stage from the oracle's state
    f.upload(fx.FIELD_VELOCITY, s.vel[0]); f.upload(fx.FIELD_COLOR, s.color); f.upload(fx.FIELD_PRESSURE, s.p)
    dt = f32(f.default_time_step())
    f.UpdateFrame(dt, 0)
    f.Advect(); f.Divergence()
    vo, _ = orc.advect(s.vel[0], s.color, dt, 0, True)
    b = orc.divergence(f.download(fx.FIELD_VELOCITY1))
    assert np.array_equal(f.download(fx.FIELD_DIVERGENCE), b)
    f.timing_read(True)
    f.Jacobi(64)
    q, k = orc.jacobi(s.p, b, 64, mode=1)
    assert np.array_equal(f.download(fx.FIELD_PRESSURE), q)
    assert f.timing_read(True).freeze_sweeps == k
    f.Project()
    assert np.array_equal(f.download(fx.FIELD_VELOCITY), orc.project(f.download(fx.FIELD_VELOCITY1), q, True))
    assert np.abs(f.download(fx.FIELD_VELOCITY1) - vo).max() < 1e-3


@pytest.mark.parametrize("dims", [(256, 256, 16), (256, 256, 41), (256, 256, 9), (256, 256, 24)])
@pytest.mark.parametrize("iters,levels", [(5, 3), (7, 6), (17, 3), (17, 9), (64, 3), (64, 12), (4, 3), (5, 4), (6, 4), (9, 8), (17, 7), (17, 11), (64, 16)])
@pytest.mark.parametrize("strip4", [1, 0])
def test_masked_strip_levels_equal_oracle(dims, iters, levels, strip4, knobs):
    """the levels right behind the dense sweep on the masked strip pipelines (k_freeze_strip4o of fx_jacobi_strip4.hip: four levels per
    launch for every cell in the octet; FREEZE_STRIP4=0: k_freeze_strip3 of fx_jacobi_stripm.hip, three; freeze nibbles carried along, both
    output buffers and the tile marks for the first tile launch) instead of tile launches: the oracle's fields and its executed-sweep
    count, whatever the number and the mix of strip launches (7 = 4 + 3, 11 = 4 + 4 + 3)"""
    if not strip4 and levels % 3:
        pytest.skip("three levels per launch")
    knobs["FLUIDX_FREEZE_DENSE_LEVELS"] = str(levels)
    knobs["FLUIDX_FREEZE_STRIP4"] = str(strip4)
    X, Y, Z = dims
    p, b = plume_like(X, Y, Z, 900 + Z + iters, amp=0.08)
    want, k = orc.jacobi(p, b, iters, mode=1)
    got, t = solve(dims, p, b, iters)
    assert np.array_equal(got, want)
    assert (t.freeze_solves, t.freeze_sweeps) == (1, k)
    launches, left, w = 0, iters - 1, levels                 # fx_schedule.cpp jacobi_freeze: fours while four levels are wanted and left behind them
    while w >= 3 and left > 3:
        lv = 4 if strip4 and w >= 4 and left > 4 else 3
        launches, left, w = launches + 1, left - lv, w - lv
    assert t.freeze_strip_launches == launches


@pytest.mark.parametrize("levels,one_copy", [(6, 1), (8, 1), (8, 0), (3, 0)])
def test_masked_strip_levels_on_frozen_and_on_restless_fields(levels, one_copy, knobs):
    """(FREEZE_DENSE_ONE=0: the dense sweep in front of a strip launch writes both copies of level 1 as it does in front of tile launches)"""
    knobs["FLUIDX_FREEZE_DENSE_LEVELS"] = str(levels)
    knobs["FLUIDX_FREEZE_DENSE_ONE"] = str(one_copy)
    dims = (256, 256, 12)
    z = np.zeros(dims[::-1], f32)
    got, t = solve(dims, z, z, 64)                       # nothing ever moves: one sweep
    assert not got.any() and t.freeze_sweeps == 1
    rng = np.random.default_rng(3)
    p = rng.standard_normal(dims[::-1]).astype(f32)
    b = (rng.standard_normal(dims[::-1]) * 0.5).astype(f32)
    want, k = orc.jacobi(p, b, 20, mode=1)               # nothing freezes early
    got, t = solve(dims, p, b, 20)
    assert np.array_equal(got, want) and t.freeze_sweeps == k


@pytest.mark.parametrize("storage", ["fp16", "fp32"])
def test_masked_strip_levels_change_no_step(storage, knobs):
    """whole steps (divergence fused into the dense sweep, three pressure and three mask buffers rotating from step to step): forced on,
    chosen by the activity count, or off -- the same fields and the same executed sweeps after 30 steps of a 256 x 256 x 48 plume"""
    dims, steps = (256, 256, 48), 30

    def run(levels):
        knobs["FLUIDX_FREEZE_DENSE_LEVELS"] = levels
        f = make(dims, jacobi_iters=64, jacobi_mode="faithful", storage=storage)
        f.timing_enable(True)
        dt = f32(f.default_time_step() * 3.0)
        for k in range(steps):
            f.UpdateFrame(dt, k % 3)
            f.Simulate(k % 3)
        f.Synchronize()
        t = f.timing_read(True)
        out = [f.download(i) for i in (fx.FIELD_VELOCITY, fx.FIELD_COLOR, fx.FIELD_PRESSURE)] + [t.freeze_sweeps, t.freeze_solves]
        f.Release()
        return out

    ref = run("0")
    assert ref[3] > 2 * steps
    for levels in ("3", "6", "4", "8", "-1"):
        got = run(levels)
        for u, v in zip(ref[:3], got[:3]):
            assert np.array_equal(u.view(np.uint8), v.view(np.uint8)), levels
        assert ref[3:] == got[3:], levels
