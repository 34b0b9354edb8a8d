"""GPU parity tests of the simulation hot path: HIP kernels (through the C ABI) vs the CPU oracle on
identical inputs.  Bar: velocity/colour within 1e-4 rel-L2 of the oracle (north_star); in practice the
kernels that contain no transcendental (divergence, Jacobi, projection) are asserted BIT-EXACT and
advection (one exp2 per voxel inside the impulse ball) to 1e-6."""
import os

import numpy as np
import pytest

import fluidx12_amd as fx
from oracle import orc

pytestmark = pytest.mark.gpu
f32 = np.float32
TOL = 1e-4          # north_star: fields within 1e-4 rel-L2 of the reference replay


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    n = np.sqrt((b ** 2).sum())
    d = np.sqrt(((a - b) ** 2).sum())
    return d / n if n > 0 else d


def rand_state(X, Y, Z, seed=0, scale=0.5):
    rng = np.random.default_rng(seed)
    vel = (rng.standard_normal((3, Z, Y, X)) * scale).astype(f32)
    col = rng.random((Z, Y, X, 4)).astype(f32)
    p = rng.standard_normal((Z, Y, X)).astype(f32)
    return vel, col, p


def make(dims, **kw):
    f = fx.Fluid()
    assert f.Init(800, 800, dims, **kw), f.last_status
    return f


DIMS = [(32, 32, 32), (64, 64, 16), (20, 20, 12), (150, 150, 6), (64, 64, 1), (36, 36, 1),
        (70, 70, 5), (130, 130, 4), (201, 201, 3)]        # rows of 2 x 35, 2 x 65 (two x tiles) and 3 x 67 cells per thread (k_*_vw)


@pytest.mark.parametrize("dims", DIMS)
@pytest.mark.parametrize("address", ["clamp", "mirror"])
def test_advect_matches_oracle(dims, address):
    X, Y, Z = dims
    vel, col, _ = rand_state(X, Y, Z, 11, scale=1.5)
    f = make(dims, advect_address=address)
    dt = f32(f.default_time_step())
    f.upload(fx.FIELD_VELOCITY, vel)
    f.upload(fx.FIELD_COLOR, col)               # parity 0 -> UpdateFrame flips: advect reads colour[!p] = this one
    f.UpdateFrame(dt, 0)
    f.Advect()
    f.Synchronize()
    vo, co = orc.advect(vel, col, dt, address=int(address == "mirror"))
    gv, gc = f.download(fx.FIELD_VELOCITY1), f.download(fx.FIELD_COLOR)
    assert rel_l2(gv, vo) < 1e-6 and rel_l2(gc, co) < 1e-6
    # outside the impulse ball there is no transcendental on the path: bit-exact
    z, y, x = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
    d2 = ((x + .5) / X - .5) ** 2 + ((y + .5) / Y - .1) ** 2 + ((z + .5) / Z - .5) ** 2
    far = d2 > (1.5 / 16) ** 2
    assert np.array_equal(gv[:, far], vo[:, far])
    assert np.array_equal(gc[far], co[far])


@pytest.mark.parametrize("dims", DIMS)
def test_divergence_jacobi_project_bit_exact(dims):
    X, Y, Z = dims
    vel, col, p = rand_state(X, Y, Z, 12)
    f = make(dims, jacobi_iters=7)
    f.upload(fx.FIELD_VELOCITY1, vel)
    f.upload(fx.FIELD_PRESSURE, p)
    f.UpdateFrame(f32(f.default_time_step()), 0)
    f.Divergence()
    b = orc.divergence(vel)
    assert np.array_equal(f.download(fx.FIELD_DIVERGENCE), b)
    f.Jacobi(7)
    q, _ = orc.jacobi(p, b, 7)
    assert np.array_equal(f.download(fx.FIELD_PRESSURE), q)
    f.Project()
    assert np.array_equal(f.download(fx.FIELD_VELOCITY), orc.project(vel, q))


@pytest.mark.parametrize("iters", [1, 2, 3, 5, 8, 20, 40])
def test_jacobi_sweep_counts(iters):
    X = 64
    _, _, p = rand_state(X, X, X, 13)
    b = np.random.default_rng(14).uniform(-1, 1, (X, X, X)).astype(f32)
    f = make((X, X, X), jacobi_iters=iters)
    f.upload(fx.FIELD_PRESSURE, p)
    f.upload(fx.FIELD_DIVERGENCE, b)
    f.Jacobi(iters)
    q, _ = orc.jacobi(p, b, iters)
    assert np.array_equal(f.download(fx.FIELD_PRESSURE), q)


@pytest.mark.parametrize("dims", [(256, 256, 200), (512, 512, 100), (256, 256, 193), (256, 256, 257), (512, 512, 97), (512, 512, 131)])
def test_default_schedule_of_large_grids_is_bit_identical(dims):
    """grids large enough for the multi-sweep strip kernels run N sweeps as fours (X = 256: k_jacobi_strip4o, X = 512: k_jacobi_strip4x)
    with remainders 5 = 3 + 2, 6 = 3 + 3, 7 = 4 + 3; every count equals N launches of one sweep bit for bit"""
    X, Y, Z = dims
    rng = np.random.default_rng(31)
    p = rng.standard_normal((Z, Y, X)).astype(f32)
    b = rng.uniform(-1, 1, (Z, Y, X)).astype(f32)
    ref = make(dims, jacobi_fuse=1)
    dut = make(dims)                                     # default schedule
    expect = {1: (1, 1), 2: (1, 2), 3: (1, 3), 4: (1, 4), 5: (2, 5), 6: (2, 6), 7: (2, 7), 8: (2, 8), 10: (3, 10), 13: (4, 13)}
    for iters, (launches, sweeps) in expect.items():
        for f in (ref, dut):
            f.upload(fx.FIELD_PRESSURE, p)
            f.upload(fx.FIELD_DIVERGENCE, b)
        dut.timing_enable(True)
        dut.timing_read(True)
        ref.Jacobi(iters)
        dut.Jacobi(iters)
        ref.Synchronize()
        dut.Synchronize()
        t = dut.timing_read(True)
        assert (t.jacobi_launches, t.jacobi_sweeps) == (launches, sweeps), (iters, t.jacobi_launches, t.jacobi_sweeps)
        assert np.array_equal(dut.download(fx.FIELD_PRESSURE), ref.download(fx.FIELD_PRESSURE)), iters


def test_jacobi_wide_strips_x512_bit_exact():
    """X = 512: the two-sweep strip kernel with two float4 per lane (k_jacobi_strip2w) == oracle, bit for bit"""
    X, Y, Z = 512, 512, 20
    _, _, p = rand_state(X, Y, Z, 19)
    b = np.random.default_rng(20).uniform(-1, 1, (Z, Y, X)).astype(f32)
    q, _ = orc.jacobi(p, b, 5)
    for fuse in (1, 2, 3):
        f = make((X, Y, Z), jacobi_iters=5, jacobi_fuse=fuse)
        f.upload(fx.FIELD_PRESSURE, p)
        f.upload(fx.FIELD_DIVERGENCE, b)
        f.Jacobi(5)                              # 2 + 2 + 1, or 3 + 2 (k_jacobi_strip3h: half-row waves with a partner mailbox)
        assert np.array_equal(f.download(fx.FIELD_PRESSURE), q), fuse


@pytest.mark.parametrize("dims", [(64, 64, 40), (128, 128, 24), (256, 256, 20)])
@pytest.mark.parametrize("fuse", [1, 2, 3, 4])
def test_jacobi_temporal_blocking_bit_exact(dims, fuse):
    """T sweeps fused in one launch (register/LDS temporal blocking) == T single sweeps == oracle, bit for bit"""
    X, Y, Z = dims
    _, _, p = rand_state(X, Y, Z, 17)
    b = np.random.default_rng(18).uniform(-1, 1, (Z, Y, X)).astype(f32)
    f = make(dims, jacobi_iters=9, jacobi_fuse=fuse)
    f.upload(fx.FIELD_PRESSURE, p)
    f.upload(fx.FIELD_DIVERGENCE, b)
    f.Jacobi(9)                                  # 9 = 4+4+1 = 3+3+3 = 2*4+1: exercises remainders
    q, _ = orc.jacobi(p, b, 9)
    assert np.array_equal(f.download(fx.FIELD_PRESSURE), q)


@pytest.mark.parametrize("dims,iters", [((150, 150, 150), 40), ((150, 150, 37), 9), ((100, 100, 9), 5), ((192, 192, 20), 8), ((160, 160, 7), 6),
                                        ((200, 200, 5), 4), ((30, 30, 30), 7), ((66, 66, 10), 6), ((8, 8, 8), 3), ((252, 252, 6), 4)])
def test_general_block_kernel_bit_exact(dims, iters):
    """rows that are no multiple of four cells or fit no strip kernel (150^3 = the reference's GI preset, Bin/FluidGI.bat) run two
    sweeps per launch in k_jacobi_blockg (X = 1 .. 4 cells per lane x up to 64 lanes, unaligned row loads, partial blocks at the
    y and z ends); odd counts end in a single sweep; == oracle, bit for bit"""
    X, Y, Z = dims
    _, _, p = rand_state(X, Y, Z, 41)
    b = np.random.default_rng(42).uniform(-1, 1, (Z, Y, X)).astype(f32)
    f = make(dims, jacobi_iters=iters)
    f.upload(fx.FIELD_PRESSURE, p)
    f.upload(fx.FIELD_DIVERGENCE, b)
    f.timing_enable(True); f.timing_read(True)
    f.Jacobi(iters)
    f.Synchronize()
    t = f.timing_read(True)
    assert t.jacobi_sweeps == iters and t.jacobi_launches == (iters + 1) // 2
    q, _ = orc.jacobi(p, b, iters)
    assert np.array_equal(f.download(fx.FIELD_PRESSURE), q)


@pytest.mark.parametrize("depth", [4, 5, 17, 33, 50, 128, 256])
def test_x256_three_sweeps_at_odd_depths_bit_exact(depth):
    """X = 256, three sweeps per launch (k_jacobi_strip3c / k_jacobi_strip3) on odd and tiny depths: chunks of unequal length, the
    pipeline's fill and drain next to both domain faces; == oracle, bit for bit"""
    dims = (256, 256, depth)
    _, _, p = rand_state(*dims, 31)
    b = np.random.default_rng(32).uniform(-1, 1, (depth, 256, 256)).astype(f32)
    f = make(dims, jacobi_iters=6, jacobi_fuse=3)
    f.upload(fx.FIELD_PRESSURE, p)
    f.upload(fx.FIELD_DIVERGENCE, b)
    f.timing_enable(True); f.timing_read(True)
    f.Jacobi(6)
    f.Synchronize()
    assert f.timing_read(True).jacobi_launches == 2
    q, _ = orc.jacobi(p, b, 6)
    assert np.array_equal(f.download(fx.FIELD_PRESSURE), q)


@pytest.mark.parametrize("kernel", ["octet", "quad"])
@pytest.mark.parametrize("depth", [2, 4, 5, 9, 17, 33, 50, 128])
def test_x256_four_sweeps_at_odd_depths_bit_exact(depth, kernel, knob):
    """X = 256, FOUR sweeps per launch on odd and tiny depths: chunks of unequal length, the pipeline's fill and drain next to both domain
    faces, the y walls in the first and last band; == oracle, bit for bit.  Both kernels: k_jacobi_strip4o (the default: eight waves of
    a workgroup, two per SIMD, over a band of 14 rows -- the inner six with every window in registers --, the last band shifted up over
    its neighbour) and k_jacobi_strip4q (four waves as a quad over 16 rows); edge rows handed over through LDS mailboxes in both"""
    knob("STRIP4_OCTET", "1" if kernel == "octet" else "0")
    rows = 256
    dims = (256, rows, depth)
    _, _, p = rand_state(*dims, 41)
    b = np.random.default_rng(42).uniform(-1, 1, (depth, rows, 256)).astype(f32)
    f = make(dims, jacobi_iters=8, jacobi_fuse=4)
    f.upload(fx.FIELD_PRESSURE, p)
    f.upload(fx.FIELD_DIVERGENCE, b)
    f.timing_enable(True); f.timing_read(True)
    f.Jacobi(8)
    f.Synchronize()
    assert f.timing_read(True).jacobi_launches == 2
    q, _ = orc.jacobi(p, b, 8)
    assert np.array_equal(f.download(fx.FIELD_PRESSURE), q)


def strip4_direct(dims, p, b, launches=2, zrange=None):
    """fx::launch_jacobi_strip4 called directly (the C++ launcher through its mangled name, device memory from torch): the C ABI only
    takes square planes (grid_x == grid_y, Fluid.cpp:201), the kernels any row count -- what a slab or a future caller may hand them"""
    import ctypes
    import re
    import subprocess
    import torch
    from fluidx12_amd import build, capi
    capi.load()
    path = os.environ.get("FLUIDX_LIB_PATH") or build.LIB       # (the library capi loaded: a lab build's switches live in ITS launchers)
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    names = re.findall(r"\b(_ZN2fx20launch_jacobi_strip4E\w+)", out)
    assert len(names) == 1, names
    fn = getattr(ctypes.CDLL(path), names[0])
    fn.restype = ctypes.c_int

    class Geom(ctypes.Structure):                  # fx_internal.h struct Geom
        _fields_ = [(n, ctypes.c_int) for n in ("X", "Y", "Zg", "z0", "nz", "H", "zlo", "zhi")]
    X, Y, Z = dims
    g = Geom(X, Y, Z, 0, Z, 0, 0, Z - 1)
    tb = torch.from_numpy(b).to("cuda")
    bufs = [torch.from_numpy(p).to("cuda"), torch.full((Z, Y, X), 7.0, dtype=torch.float32, device="cuda")]
    vp = ctypes.c_void_p
    z0, z1 = zrange or (0, Z)
    for k in range(launches):
        rc = fn(ctypes.byref(g), vp(bufs[k & 1].data_ptr()), vp(tb.data_ptr()), vp(bufs[(k + 1) & 1].data_ptr()), ctypes.c_int(z0), ctypes.c_int(z1), vp(0))
        if rc != 0:
            return rc, None
    torch.cuda.synchronize()
    return 0, bufs[launches & 1].cpu().numpy()


def explain(got, want):
    bad = np.argwhere(got != want)
    return (len(bad), bad[:5].tolist(), "x halves", np.unique(bad[:, 2] // 256).tolist(), "rows", np.unique(bad[:, 1])[:24].tolist(), "planes", np.unique(bad[:, 0])[:24].tolist())


@pytest.mark.parametrize("dims", [(512, 512, 2), (512, 512, 5), (512, 512, 9), (512, 512, 21), (512, 512, 40)])
def test_x512_four_sweeps_bit_exact(dims):
    """X = 512, FOUR sweeps per launch (k_jacobi_strip4x: the octet's pipeline on half-row waves, the x cut inside the workgroup -- input
    cells across it fetched with the plane, the cells of levels 1..3 through 16-byte LDS slots under the edge rows' counters): odd and tiny
    depths (runs of unequal length, pieces that cross into the next band, fill and drain at both faces; 512 rows = 85 bands of six + a
    shifted last band, and the band before it shifted as well: its halo would cross the last row), 8 = 4 + 4 sweeps; == oracle bit for
    bit, twice (two writers of a shared row racing would not repeat)"""
    X, Y, Z = dims
    _, _, p = rand_state(X, Y, Z, 61)
    b = np.random.default_rng(62).uniform(-1, 1, (Z, Y, X)).astype(f32)
    q, _ = orc.jacobi(p, b, 8)
    f = make(dims, jacobi_iters=8, jacobi_fuse=4)
    for _ in range(2):
        f.upload(fx.FIELD_PRESSURE, p)
        f.upload(fx.FIELD_DIVERGENCE, b)
        f.timing_enable(True); f.timing_read(True)
        f.Jacobi(8)
        f.Synchronize()
        assert f.timing_read(True).jacobi_launches == 2
        got = f.download(fx.FIELD_PRESSURE)
        assert np.array_equal(got, q), explain(got, q)


@pytest.mark.parametrize("dims", [(512, 6, 21), (512, 9, 12), (512, 10, 12), (512, 11, 12), (512, 12, 12), (512, 13, 33), (512, 14, 12), (512, 20, 12),
                                  (512, 31, 17), (512, 100, 50), (512, 256, 64), (512, 510, 11), (512, 511, 11), (512, 513, 11)])
def test_x512_four_sweeps_on_any_row_count_bit_exact(dims):
    """the half-row octet's bands of six rows on row counts they tile and do not tile (the launcher called directly: the C ABI takes square
    planes only); Y = 7, 8 fit no placement and are refused"""
    X, Y, Z = dims
    _, _, p = rand_state(X, Y, Z, 65)
    b = np.random.default_rng(66).uniform(-1, 1, (Z, Y, X)).astype(f32)
    q, _ = orc.jacobi(p, b, 8)
    for _ in range(2):
        rc, got = strip4_direct(dims, p, b)
        assert rc == 0
        assert np.array_equal(got, q), explain(got, q)
    assert strip4_direct((512, 7, 4), p[:4, :7].copy(), b[:4, :7].copy())[0] != 0


def test_x512_four_sweeps_beyond_the_infinity_cache_bit_exact():
    """from 40 M cells (p + b no longer fit the Infinity Cache) k_jacobi_strip4x<NT> writes its output with non-temporal stores:
    512 x 512 x 160 (42 M cells), 8 = 4 + 4 sweeps == oracle bit for bit"""
    dims = (512, 512, 160)
    rng = np.random.default_rng(69)
    p = rng.standard_normal((160, 512, 512)).astype(f32)
    b = rng.uniform(-1, 1, (160, 512, 512)).astype(f32)
    q, _ = orc.jacobi(p, b, 8)
    f = make(dims, jacobi_iters=8, jacobi_fuse=4)
    f.upload(fx.FIELD_PRESSURE, p)
    f.upload(fx.FIELD_DIVERGENCE, b)
    f.Jacobi(8)
    got = f.download(fx.FIELD_PRESSURE)
    assert np.array_equal(got, q), explain(got, q)


@pytest.mark.parametrize("zrange", [(5, 6), (0, 1), (11, 12), (4, 6), (3, 11)])
def test_x512_four_sweeps_on_a_range_of_planes(zrange):
    """a launch over a few planes inside a deeper grid (what a slab rank's shrinking rounds hand the kernel): 86 bands x 1 plane are runs of
    whole bands (at most seven pieces per run); the planes of the range == four sweeps of the oracle, the planes outside it untouched"""
    dims = (512, 512, 12)
    _, _, p = rand_state(*dims, 67)
    b = np.random.default_rng(68).uniform(-1, 1, (12, 512, 512)).astype(f32)
    q, _ = orc.jacobi(p, b, 4)
    rc, got = strip4_direct(dims, p, b, launches=1, zrange=zrange)
    assert rc == 0
    z0, z1 = zrange
    assert np.array_equal(got[z0:z1], q[z0:z1]), explain(got[z0:z1], q[z0:z1])
    out = np.ones(12, bool); out[z0:z1] = False
    assert np.all(got[out] == f32(7.0))


@pytest.mark.parametrize("wgs", [1, 7, 64, 200, 256, 300])
def test_x512_four_sweeps_any_number_of_runs(wgs, knob):
    """the band-planes of a launch are cut into one contiguous run per workgroup; any number of runs gives the same bits"""
    knob("STRIP4X_WGS", str(wgs))
    dims = (512, 100, 30)
    _, _, p = rand_state(*dims, 63)
    b = np.random.default_rng(64).uniform(-1, 1, (30, 100, 512)).astype(f32)
    q, _ = orc.jacobi(p, b, 4)
    rc, got = strip4_direct(dims, p, b, launches=1)
    assert rc == 0 and np.array_equal(got, q), explain(got, q)


@pytest.mark.parametrize("dims", [(68, 17, 9), (100, 30, 12), (132, 20, 8), (160, 45, 12), (192, 33, 21), (200, 14, 12), (224, 29, 7), (252, 18, 5)])
def test_rows_shorter_than_256_cells_four_sweeps_bit_exact(dims):
    """rows of 68..252 cells (whole quads) as ONE tile of k_jacobi_strip4t with its upper lanes switched off for the whole walk: the
    row's last lane finds no source for its right-hand neighbour and keeps its own cell -- the wall -- as lane 63 of a full row does;
    8 = 4 + 4 sweeps == oracle bit for bit, twice.  The launcher directly: any Y."""
    X, Y, Z = dims
    _, _, p = rand_state(X, Y, Z, 81)
    b = np.random.default_rng(82).uniform(-1, 1, (Z, Y, X)).astype(f32)
    q, _ = orc.jacobi(p, b, 8)
    for _ in range(2):
        rc, got = strip4_direct(dims, p, b)
        assert rc == 0
        assert np.array_equal(got, q), explain(got, q)


@pytest.mark.parametrize("dims", [(260, 17, 9), (264, 30, 12), (320, 20, 21), (384, 45, 12), (500, 14, 12), (504, 33, 7), (508, 18, 5), (516, 29, 12),
                                  (752, 17, 6), (756, 40, 9), (1000, 19, 6), (1024, 31, 10), (2048, 17, 5)])
def test_any_row_length_four_sweeps_bit_exact(dims):
    """rows longer than 256 cells that are neither 256 nor 512 ("any other row length falls to the generic paths"):
    k_jacobi_strip4t, the octet on x tiles of 256 cells 248 apart -- a tile's outermost lane on a side that is no wall is its own recomputed
    halo and stores nothing.  Tile counts 2..9, last tiles that keep 2 to 62 lanes, row counts that do and do not tile the bands, odd depths;
    8 = 4 + 4 sweeps == oracle bit for bit, twice (two writers of one cell racing would not repeat).  The launcher directly: any Y."""
    X, Y, Z = dims
    _, _, p = rand_state(X, Y, Z, 71)
    b = np.random.default_rng(72).uniform(-1, 1, (Z, Y, X)).astype(f32)
    q, _ = orc.jacobi(p, b, 8)
    for _ in range(2):
        rc, got = strip4_direct(dims, p, b)
        assert rc == 0
        assert np.array_equal(got, q), explain(got, q)


@pytest.mark.parametrize("dims,piece_min", [((4096, 406, 10), 8), ((4096, 406, 21), 5), ((1024, 1246, 9), 4), ((2048, 700, 13), 3)])
def test_any_row_length_one_piece_per_workgroup(dims, piece_min, knob):
    """more (tile, band) pairs than CUs: the launcher hands every workgroup one piece (a band's planes of one z chunk) instead of a run --
    493 / 445 / 450 bands, one to four chunks, depths the chunks do not divide (the cuts then fall a few planes beside the piece boundaries:
    any cut gives the same bits); STRIP4T_PIECES (lab builds) lowers the 64 planes a piece must have so that the case fits a test"""
    knob("STRIP4T_PIECES", str(piece_min))
    X, Y, Z = dims
    _, _, p = rand_state(X, Y, Z, 75)
    b = np.random.default_rng(76).uniform(-1, 1, (Z, Y, X)).astype(f32)
    q, _ = orc.jacobi(p, b, 4)
    rc, got = strip4_direct(dims, p, b, launches=1)
    assert rc == 0 and np.array_equal(got, q), explain(got, q)


def test_any_row_length_one_piece_per_workgroup_at_full_depth():
    """... with the shipped thresholds: 1024 x 1246 x 64 (5 tiles x 89 bands = 445 pieces of 64 planes, 1.7 rounds of 256), four sweeps ==
    oracle bit for bit"""
    dims = (1024, 1246, 64)
    rng = np.random.default_rng(77)
    p = rng.standard_normal((64, 1246, 1024)).astype(f32)
    b = rng.uniform(-1, 1, (64, 1246, 1024)).astype(f32)
    q, _ = orc.jacobi(p, b, 4)
    rc, got = strip4_direct(dims, p, b, launches=1)
    assert rc == 0 and np.array_equal(got, q), explain(got, q)


@pytest.mark.parametrize("zrange", [(5, 6), (0, 1), (11, 12), (3, 11)])
def test_any_row_length_four_sweeps_on_a_range_of_planes(zrange):
    """... over a few planes inside a deeper grid: the planes of the range == four sweeps of the oracle, the others untouched"""
    dims = (384, 40, 12)
    _, _, p = rand_state(*dims, 73)
    b = np.random.default_rng(74).uniform(-1, 1, (12, 40, 384)).astype(f32)
    q, _ = orc.jacobi(p, b, 4)
    rc, got = strip4_direct(dims, p, b, launches=1, zrange=zrange)
    assert rc == 0
    z0, z1 = zrange
    assert np.array_equal(got[z0:z1], q[z0:z1]), explain(got[z0:z1], q[z0:z1])
    out = np.ones(12, bool); out[z0:z1] = False
    assert np.all(got[out] == f32(7.0))


@pytest.mark.parametrize("rows", [14, 15, 16, 17, 18, 19, 27, 28, 29, 30, 31, 44, 45, 58, 100, 128, 240, 254])
def test_x256_four_sweeps_on_rows_that_do_not_tile_the_bands_bit_exact(rows):
    """the octet's bands of 14 rows on row counts they do not tile (the launcher called directly: the C ABI takes square planes only).
    Y % 14 = 1 or 2 (29, 30, 44, 58, 128, 240): the lower halo of the second-to-last band would reach beyond the last row -- that band is
    shifted up like the last one (octet_band_y; round 5 computed level-l rows behind the wall from clamped loads and let two workgroups
    store different bits to one address: ADVICE round 5).  Y = 15, 16 fit no placement (the last band's upper halo would cross the first
    row) and are refused.  == oracle bit for bit, twice (a race between the two writers of a shared row would not repeat)"""
    depth = 21
    dims = (256, rows, depth)
    _, _, p = rand_state(*dims, 51)
    b = np.random.default_rng(52).uniform(-1, 1, (depth, rows, 256)).astype(f32)
    q, _ = orc.jacobi(p, b, 8)
    for _ in range(2):
        rc, got = strip4_direct(dims, p, b)
        if rows in (15, 16):
            assert rc != 0
            return
        assert rc == 0
        assert np.array_equal(got, q), explain(got, q)


@pytest.mark.parametrize("kernel", ["octet", "quad"])
def test_x256_default_schedule_runs_fours_bit_exact(kernel, knob):
    """the default schedule at 256^3-class sizes: 40 sweeps = ten launches of four sweeps (k_jacobi_strip4o, or k_jacobi_strip4q with
    STRIP4_OCTET=0); 256 x 256 x 200 against the oracle, and odd sweep counts (remainders of 5 / 6 / 7 as 3 + 2 / 3 + 3 / 4 + 3)"""
    knob("STRIP4_OCTET", "1" if kernel == "octet" else "0")
    dims = (256, 256, 200)
    _, _, p = rand_state(*dims, 43)
    b = np.random.default_rng(44).uniform(-1, 1, (200, 256, 256)).astype(f32)
    for iters, launches in ((40, 10), (13, 4), (14, 4), (15, 4), (9, 3)):
        f = make(dims, jacobi_iters=iters)
        f.upload(fx.FIELD_PRESSURE, p)
        f.upload(fx.FIELD_DIVERGENCE, b)
        f.timing_enable(True); f.timing_read(True)
        f.Jacobi(iters)
        f.Synchronize()
        t = f.timing_read(True)
        assert t.jacobi_sweeps == iters and t.jacobi_launches == launches, (iters, t.jacobi_launches)
        q, _ = orc.jacobi(p, b, iters)
        assert np.array_equal(f.download(fx.FIELD_PRESSURE), q), iters
        f.Release()


@pytest.mark.parametrize("overlap", [0, 2])
def test_x256_three_sweeps_in_slabs(overlap):
    """the same kernels on the shrinking ranges of z-slabs (halo planes included, ranges that start and end inside the slab)"""
    from test_gpu_slabs import run_single, run_slabs, gather
    dims = (256, 256, 400)
    ref = run_single(dims, 2, jacobi_iters=19, jacobi_fuse=1)
    fl = run_slabs(dims, 2, 2, jacobi_iters=19, halo_jacobi=8, halo_advect=8, overlap=overlap)
    assert np.array_equal(gather(fl, fx.FIELD_PRESSURE, 0), ref.download(fx.FIELD_PRESSURE))
    assert np.array_equal(gather(fl, fx.FIELD_VELOCITY, 1), ref.download(fx.FIELD_VELOCITY))


@pytest.mark.parametrize("dims,iters", [((128, 128, 128), 9), ((128, 128, 30), 7), ((128, 128, 5), 4), ((128, 128, 128), 40)])
def test_x128_default_schedule_block_kernel_bit_exact(dims, iters):
    """X = 128 (BASELINE configs[1], the reference's default grid): the default schedule runs two sweeps per launch in
    k_jacobi_block2 (one 4 x 4-row block per wave), odd counts end in a single sweep; == oracle, bit for bit, including the
    partial blocks of depths that are no multiple of four"""
    X, Y, Z = dims
    _, _, p = rand_state(X, Y, Z, 23)
    b = np.random.default_rng(24).uniform(-1, 1, (Z, Y, X)).astype(f32)
    f = make(dims, jacobi_iters=iters)
    f.upload(fx.FIELD_PRESSURE, p)
    f.upload(fx.FIELD_DIVERGENCE, b)
    f.timing_enable(True); f.timing_read(True)
    f.Jacobi(iters)
    f.Synchronize()
    t = f.timing_read(True)
    assert t.jacobi_sweeps == iters and t.jacobi_launches == (iters + 1) // 2
    q, _ = orc.jacobi(p, b, iters)
    assert np.array_equal(f.download(fx.FIELD_PRESSURE), q)


def test_jacobi_faithful_mode_matches_oracle():
    X = 32
    s = orc.Sim(X, X, X, iters=64, mode=1)
    f = make((X, X, X), jacobi_iters=64, jacobi_mode="faithful")
    for _ in range(3):
        s.step()
        f.UpdateFrame(f32(f.default_time_step()), 0)
        f.Simulate(0)
    f.Synchronize()
    assert rel_l2(f.download(fx.FIELD_PRESSURE), s.p) < 1e-5
    assert rel_l2(f.download(fx.FIELD_VELOCITY), s.velocity) < TOL


@pytest.mark.parametrize("dims,iters,steps", [((64, 64, 1), 20, 8), ((32, 32, 32), 40, 8), ((48, 48, 24), 40, 4)])
@pytest.mark.parametrize("address", ["clamp", "mirror"])
def test_rollout_matches_oracle(dims, iters, steps, address):
    """config 1 (64^2, 20 sweeps) and small 3D grids: <= 8 steps from the zero state (the flow is chaotic,
    rounding noise grows ~1.7x per step -- SURVEY.md hard part 3)."""
    X, Y, Z = dims
    s = orc.Sim(X, Y, Z, iters=iters, address=int(address == "mirror"))
    f = make(dims, jacobi_iters=iters, advect_address=address)
    for k in range(steps):
        s.step()
        f.UpdateFrame(f32(f.default_time_step()), k % 3)
        f.Simulate(k % 3)
    f.Synchronize()
    assert rel_l2(f.download(fx.FIELD_VELOCITY), s.velocity) < TOL
    assert rel_l2(f.download(fx.FIELD_COLOR), s.color) < TOL
    assert rel_l2(f.download(fx.FIELD_PRESSURE), s.p) < TOL
    assert f.frame_info().frame_parity == s.parity


def test_fp16_storage_rollout():
    X = 32
    s = orc.Sim(X, X, X, iters=40, half=True)
    f = make((X, X, X), jacobi_iters=40, storage="fp16")
    for k in range(4):
        s.step()
        f.UpdateFrame(f32(f.default_time_step()), 0)
        f.Simulate(0)
    f.Synchronize()
    gv, gc = f.download(fx.FIELD_VELOCITY), f.download(fx.FIELD_COLOR)
    # fp32 arithmetic, then a separate RNE to binary16 on both sides: only an exp2 ulp flipping a stored rounding can differ
    assert rel_l2(gv, s.velocity) < 1e-4 and rel_l2(gc, s.color) < 1e-4
    # stored values are exactly representable in binary16
    assert np.array_equal(gv.astype(np.float16).astype(f32), gv)


def test_fp16_single_step_kernels():
    X = 32
    vel, col, p = rand_state(X, X, X, 15)
    vel = vel.astype(np.float16).astype(f32); col = col.astype(np.float16).astype(f32)
    f = make((X, X, X), storage="fp16")
    dt = f32(f.default_time_step())
    f.upload(fx.FIELD_VELOCITY, vel); f.upload(fx.FIELD_COLOR, col)
    f.UpdateFrame(dt, 0)
    f.Advect()
    vo, co = orc.advect(vel, col, dt, half=True)
    gv, gc = f.download(fx.FIELD_VELOCITY1), f.download(fx.FIELD_COLOR)
    # same fp32 arithmetic then RNE to half: at most a rare 1-ulp(half) flip from the exp2 difference
    assert np.mean(gv != vo) < 1e-4 and np.mean(gc != co) < 1e-4
    f.upload(fx.FIELD_VELOCITY1, vo); f.upload(fx.FIELD_PRESSURE, p)
    f.Project()
    assert np.array_equal(f.download(fx.FIELD_VELOCITY), orc.project(vo, p, half=True))


def test_fp16_store_rounds_the_fp32_result_not_the_exact_product():
    """dt = 1/4 makes the attenuation 0.95f, and u * 0.95f lands on binary16 ties for many u: the fp32 product is rounded to
    fp32 FIRST and then to binary16 (RNE, tie to even).  A fused multiply-convert (v_fma_mixlo_f16) rounds the exact
    product once and goes the other way; found by the fuzz soak on an 8 x 8 x 29 grid."""
    dims = (8, 8, 29)
    rng = np.random.default_rng(794)
    vel = (rng.standard_normal((3, 29, 8, 8)) * 3).astype(np.float16).astype(f32)
    col = rng.random((29, 8, 8, 4)).astype(np.float16).astype(f32)
    f = make(dims, storage="fp16")
    dt = f32(f.default_time_step())
    assert dt == f32(0.25)
    f.upload(fx.FIELD_VELOCITY, vel); f.upload(fx.FIELD_COLOR, col)
    f.UpdateFrame(dt, 0)
    f.Advect()
    f.Synchronize()
    vo, co = orc.advect(vel, col, dt, half=True)
    z, y, x = np.meshgrid(np.arange(29), np.arange(8), np.arange(8), indexing="ij")
    far = ((x + .5) / 8 - .5) ** 2 + ((y + .5) / 8 - .1) ** 2 + ((z + .5) / 29 - .5) ** 2 > (1.5 / 16) ** 2
    gv, gc = f.download(fx.FIELD_VELOCITY1), f.download(fx.FIELD_COLOR)
    assert np.array_equal(gv[:, far], vo[:, far]) and np.array_equal(gc[far], co[far])       # no transcendental there: bit-exact
    # (with the fused conversion this very state differed in 52 velocity and 45 colour values, all outside the impulse ball)


def test_paused_step_copies_velocity():
    X = 16
    vel, col, p = rand_state(X, X, X, 16)
    f = make((X, X, X))
    f.upload(fx.FIELD_VELOCITY, vel); f.upload(fx.FIELD_COLOR, col); f.upload(fx.FIELD_PRESSURE, p)
    f.UpdateFrame(0.0, 0)                        # dt = 0: parity does not flip (Fluid.cpp:345)
    f.Simulate(0)
    f.Synchronize()
    assert np.array_equal(f.download(fx.FIELD_VELOCITY), vel)      # CSProject3D.hlsl:88,112
    assert np.array_equal(f.download(fx.FIELD_PRESSURE), p)
    assert f.frame_info().frame_parity == 0


def test_errors_and_call_order():
    f = fx.Fluid()
    assert f.Init(800, 800, (32, 16, 8)) is False                  # x != y, Fluid.cpp:201
    f = make((16, 16, 16))
    with pytest.raises(fx.FluidxError):
        f.Simulate(0)                                              # UpdateFrame first
    with pytest.raises(fx.FluidxError):
        f.UpdateFrame(0.1, 3)                                      # frameIndex < FrameCount
    with pytest.raises(ValueError):
        f.upload(fx.FIELD_PRESSURE, np.zeros((4, 4, 4), f32))


@pytest.mark.parametrize("dims,iters,launches", [((384, 384, 40), 40, 10), ((320, 320, 24), 23, 8), ((264, 264, 30), 6, 3), ((192, 192, 96), 40, 10), ((252, 252, 56), 9, 3)])
def test_any_row_length_full_step_against_oracle(dims, iters, launches):
    """a grid whose rows are neither 256 nor 512 cells through one whole step, stage by stage against the oracle: the default schedule takes
    FOUR sweeps per launch on x tiles of the octet (k_jacobi_strip4t; 23 sweeps = 5 x 4 + 1 + 1 + 1: these rows have no three- or
    two-sweep kernel; rows of 192 / 252 cells: one tile with its upper lanes switched off, from 3.1 M cells, 9 sweeps = 4 + 4 + 1), advection / divergence / projection their general kernels; then fx_simulate as a whole against the staged run"""
    X, Y, Z = dims
    rng = np.random.default_rng(384)
    vel = (rng.random((3, Z, Y, X), dtype=f32) - f32(0.5)) * f32(4.0)
    col = rng.random((Z, Y, X, 4), dtype=f32)
    p = rng.standard_normal((Z, Y, X)).astype(f32)
    f = make(dims, jacobi_iters=iters)
    dt = f32(f.default_time_step())
    f.upload(fx.FIELD_VELOCITY, vel); f.upload(fx.FIELD_COLOR, col); f.upload(fx.FIELD_PRESSURE, p)
    f.UpdateFrame(dt, 0)
    f.Advect()
    gv, gc = f.download(fx.FIELD_VELOCITY1), f.download(fx.FIELD_COLOR)
    vo, co = orc.advect(vel, col, dt)
    assert rel_l2(gv, vo) < 1e-6 and rel_l2(gc, co) < 1e-6
    f.Divergence()
    b = orc.divergence(gv)
    assert np.array_equal(f.download(fx.FIELD_DIVERGENCE), b)
    f.timing_enable(True); f.timing_read(True)
    f.Jacobi(iters)
    f.Synchronize()
    t = f.timing_read(True)
    assert t.jacobi_sweeps == iters and t.jacobi_launches == launches, (t.jacobi_sweeps, t.jacobi_launches)
    q, _ = orc.jacobi(p, b, iters)
    got = f.download(fx.FIELD_PRESSURE)
    assert np.array_equal(got, q), explain(got, q)
    f.Project()
    want = orc.project(gv, q)
    assert np.array_equal(f.download(fx.FIELD_VELOCITY), want)
    g = make(dims, jacobi_iters=iters)
    g.upload(fx.FIELD_VELOCITY, vel); g.upload(fx.FIELD_COLOR, col); g.upload(fx.FIELD_PRESSURE, p)
    g.UpdateFrame(dt, 0)
    g.Simulate(0)
    g.Synchronize()
    assert np.array_equal(g.download(fx.FIELD_VELOCITY), want) and np.array_equal(g.download(fx.FIELD_COLOR), gc)


# ---- full BASELINE sizes: size-independent properties --------------------------------------------------
def test_x512_full_step_against_oracle():
    """BASELINE configs[3]'s kernels (X = 512, 80 sweeps: k_jacobi_strip3h in threes + k_jacobi_strip2h for the remainder) through one
    whole step on a 512 x 512 x 100 grid -- thick enough for the default schedule to take the three-sweep kernel -- stage by stage
    against the oracle on the same inputs, then fx_simulate as a whole against the staged run"""
    dims = (512, 512, 100)
    X, Y, Z = dims
    rng = np.random.default_rng(512)
    vel = (rng.random((3, Z, Y, X), dtype=f32) - f32(0.5)) * f32(4.0)       # reach up to 4 cells in z, 4 in x / y
    col = rng.random((Z, Y, X, 4), dtype=f32)
    p = rng.standard_normal((Z, Y, X)).astype(f32)
    f = make(dims, jacobi_iters=80)
    dt = f32(f.default_time_step())
    f.upload(fx.FIELD_VELOCITY, vel); f.upload(fx.FIELD_COLOR, col); f.upload(fx.FIELD_PRESSURE, p)
    f.UpdateFrame(dt, 0)
    f.Advect()
    gv, gc = f.download(fx.FIELD_VELOCITY1), f.download(fx.FIELD_COLOR)
    vo, co = orc.advect(vel, col, dt)
    assert rel_l2(gv, vo) < 1e-6 and rel_l2(gc, co) < 1e-6
    z, y, x = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij", sparse=True)
    far = ((x + .5) / X - .5) ** 2 + ((y + .5) / Y - .1) ** 2 + ((z + .5) / Z - .5) ** 2 > (1.5 / 16) ** 2
    assert np.array_equal(gv[:, far], vo[:, far]) and np.array_equal(gc[far], co[far])       # no transcendental there: bit-exact
    f.Divergence()
    b = orc.divergence(gv)
    assert np.array_equal(f.download(fx.FIELD_DIVERGENCE), b)
    f.timing_enable(True); f.timing_read(True)
    f.Jacobi(80)
    f.Synchronize()
    t = f.timing_read(True)
    assert t.jacobi_sweeps == 80 and t.jacobi_launches == 20 and t.jacobi_main_sweeps == 80      # 20 x 4 (k_jacobi_strip4x)
    q, _ = orc.jacobi(p, b, 80)
    assert np.array_equal(f.download(fx.FIELD_PRESSURE), q)
    f.Project()
    want = orc.project(gv, q)
    assert np.array_equal(f.download(fx.FIELD_VELOCITY), want)
    # the same step through fx_simulate
    g = make(dims, jacobi_iters=80)
    g.upload(fx.FIELD_VELOCITY, vel); g.upload(fx.FIELD_COLOR, col); g.upload(fx.FIELD_PRESSURE, p)
    g.UpdateFrame(dt, 0)
    g.Simulate(0)
    g.Synchronize()
    assert np.array_equal(g.download(fx.FIELD_VELOCITY), want) and np.array_equal(g.download(fx.FIELD_COLOR), gc)
    assert np.array_equal(g.download(fx.FIELD_PRESSURE), q)


@pytest.mark.parametrize("storage,address", [("fp32", "clamp"), ("fp32", "mirror"), ("fp16", "clamp"), ("fp16", "mirror")])
def test_x256_full_step_against_oracle(storage, address):
    """The headline grid itself (BASELINE configs[2], and configs[4] with fp16 storage): one whole step at 256^3 from a DEVELOPED state
    -- 48 steps of the plume plus a patch of fast random flow, so that k_advect_lds serves lanes from its LDS tile and lanes through its
    gather path -- stage by stage against the oracle on the same inputs: advection (k_advect_lds), divergence, 40 sweeps in the default
    schedule (10 x k_jacobi_strip4o, the octet), projection; then fx_simulate as a whole against the staged run."""
    dims = (256, 256, 256)
    X, Y, Z = dims
    half = storage == "fp16"
    amode = int(address == "mirror")
    f = make(dims, storage=storage, jacobi_iters=40, advect_address=address)
    dt = f32(f.default_time_step())
    for k in range(48):
        f.UpdateFrame(dt, k % 3)
        f.Simulate(k % 3)
    vel, col, p = f.download(fx.FIELD_VELOCITY), f.download(fx.FIELD_COLOR), f.download(fx.FIELD_PRESSURE)
    assert np.abs(vel).max() > 0.05                                     # the plume moves
    rng = np.random.default_rng(256 + amode + 2 * half)
    vel[:, 150:200, 100:180, 60:200] += (rng.random((3, 50, 80, 140), dtype=f32) - f32(0.5)) * f32(3.0)     # back-traces of up to 3 cells
    if half:
        vel = vel.astype(np.float16).astype(f32)
    reach = np.abs(vel).max(axis=0) * dt * X
    assert 0.005 < (reach >= 1.0).mean() < 0.5                          # both tap sources are exercised
    f.upload(fx.FIELD_VELOCITY, vel); f.upload(fx.FIELD_COLOR, col); f.upload(fx.FIELD_PRESSURE, p)
    f.UpdateFrame(dt, 0)
    f.Advect()
    gv, gc = f.download(fx.FIELD_VELOCITY1), f.download(fx.FIELD_COLOR)
    vo, co = orc.advect(vel, col, dt, address=amode, half=half)
    assert rel_l2(gv, vo) < (2e-4 if half else 1e-6) and rel_l2(gc, co) < (2e-4 if half else 1e-6)       # fp16: an exp2 ulp can flip a binary16 rounding
    z, y, x = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij", sparse=True)
    far = ((x + .5) / X - .5) ** 2 + ((y + .5) / Y - .1) ** 2 + ((z + .5) / Z - .5) ** 2 > (1.5 / 16) ** 2
    assert np.array_equal(gv[:, far], vo[:, far]) and np.array_equal(gc[far], co[far])       # no transcendental there: bit-exact
    del vo, co, far
    f.Divergence()
    b = orc.divergence(gv)
    assert np.array_equal(f.download(fx.FIELD_DIVERGENCE), b)
    f.timing_enable(True); f.timing_read(True)
    f.Jacobi(40)
    f.Synchronize()
    t = f.timing_read(True)
    f.timing_enable(False)
    assert t.jacobi_sweeps == 40 and t.jacobi_launches == 10 and t.jacobi_main_sweeps == 40      # 10 x 4 (k_jacobi_strip4o)
    q, _ = orc.jacobi(p, b, 40)
    assert np.array_equal(f.download(fx.FIELD_PRESSURE), q)
    f.Project()
    want = orc.project(gv, q, half)
    assert np.array_equal(f.download(fx.FIELD_VELOCITY), want)
    # the same step through fx_simulate
    g = make(dims, storage=storage, jacobi_iters=40, advect_address=address)
    g.upload(fx.FIELD_VELOCITY, vel); g.upload(fx.FIELD_COLOR, col); g.upload(fx.FIELD_PRESSURE, p)
    g.UpdateFrame(dt, 0)
    g.Simulate(0)
    g.Synchronize()
    assert np.array_equal(g.download(fx.FIELD_VELOCITY), want) and np.array_equal(g.download(fx.FIELD_COLOR), gc)
    assert np.array_equal(g.download(fx.FIELD_PRESSURE), q)


@pytest.mark.parametrize("X", [128, 256, 512])
def test_full_size_properties(X):
    f = make((X, X, X), jacobi_iters=80 if X == 512 else 40)
    dt = f32(f.default_time_step())
    for k in range(2):
        f.UpdateFrame(dt, 0)
        f.Simulate(0)
    f.Synchronize()
    u = f.download(fx.FIELD_VELOCITY)
    c = f.download(fx.FIELD_COLOR)
    assert np.isfinite(u).all() and np.isfinite(c).all()
    ux, uy, uz = u
    scale = np.abs(u).max()
    assert scale > 0
    # 180-degree rotational symmetry about the vertical axis through (0.5, ., 0.5)
    assert np.abs(ux + ux[::-1, :, ::-1]).max() < 2e-4 * scale
    assert np.abs(uy - uy[::-1, :, ::-1]).max() < 2e-4 * scale
    assert np.abs(c[..., 3] - c[::-1, :, ::-1, 3]).max() < 1e-4
    assert (c >= 0).all() and (c <= 1).all()                       # saturate + dissipation
    # linearity of the sweep: J(p1 + p2, b1 + b2) == J(p1, b1) + J(p2, b2) up to rounding
    rng = np.random.default_rng(21)
    p1, p2, b1, b2 = (rng.standard_normal((X, X, X)).astype(f32) for _ in range(4))

    def J(p, b):
        f.upload(fx.FIELD_PRESSURE, p); f.upload(fx.FIELD_DIVERGENCE, b)
        f.Jacobi(8)
        return f.download(fx.FIELD_PRESSURE)
    lhs, rhs = J(p1 + p2, b1 + b2), J(p1, b1) + J(p2, b2)
    assert rel_l2(lhs, rhs) < 1e-6
    # a slab of the full-size sweep agrees bit-for-bit with the oracle run on that slab's dependency cone
    zs = slice(0, 12)
    q, _ = orc.jacobi(p1[:12 + 8], b1[:12 + 8], 8)
    assert np.array_equal(J(p1, b1)[0:12], q[0:12])


@pytest.mark.parametrize("dims,storage,address", [((64, 64, 32), "fp32", "clamp"), ((32, 32, 64), "fp16", "mirror"),
                                                   ((128, 128, 1), "fp32", "clamp"), ((32, 32, 16), "fp32", "mirror"),
                                                   ((256, 256, 16), "fp32", "clamp")])
def test_advect_fast_path_bit_identical(dims, storage, address, knob):
    """k_advect_fast (power-of-two grids: reciprocal multiplies, 32-bit tap offsets, shifts) against the general kernel"""
    import fluidx12_amd as fx

    def run(fast):
        knob("ADVECT_FAST", "1" if fast else "0")
        f = fx.Fluid()
        assert f.Init(320, 240, dims, storage=storage, advect_address=address, jacobi_iters=6)
        for k in range(5):
            f.UpdateFrame(np.float32(f.default_time_step()), k % 3)
            f.Simulate(k % 3)
        f.Synchronize()
        return f.download(fx.FIELD_VELOCITY), f.download(fx.FIELD_COLOR)

    v0, c0 = run(False)
    v1, c1 = run(True)
    assert np.abs(c0).max() > 0
    assert np.array_equal(v0.view(np.uint32), v1.view(np.uint32)) and np.array_equal(c0.view(np.uint32), c1.view(np.uint32))


@pytest.mark.parametrize("storage", ["fp32", "fp16"])
@pytest.mark.parametrize("dims,address,scale", [((64, 64, 64), "clamp", 0.2), ((64, 64, 64), "mirror", 3.0), ((128, 128, 32), "clamp", 1.0),
                                                 ((256, 256, 16), "mirror", 0.4), ((64, 64, 16), "clamp", 12.0),
                                                 ((150, 150, 24), "clamp", 1.0), ((150, 150, 20), "mirror", 3.0), ((100, 100, 33), "clamp", 0.3),
                                                 ((72, 72, 16), "mirror", 12.0)])
def test_advect_lds_path_bit_identical(dims, address, scale, storage, knob):
    """k_advect_lds (taps from an LDS-staged 66 x 10 x 3-plane window, global gathers for the waves that trace further) against
    k_advect_fast and the oracle: random velocities from well inside the window (scale 0.2: every wave on the LDS path) to far
    outside (12: every wave on the gather path), both addressing modes, grid borders in every direction; and extents that are no
    power of two (150: the reference's GI preset, Bin/FluidGI.bat:1 -- tiles with lanes and whole waves beyond the grid)"""
    X, Y, Z = dims
    rng = np.random.default_rng(77)
    vel = (rng.standard_normal((3, Z, Y, X)) * scale).astype(f32)
    vel[:, :, : Y // 2] *= f32(0.05)                       # half of the rows trace less than a cell: mixed waves inside one workgroup
    col = rng.random((Z, Y, X, 4)).astype(f32)
    half = storage == "fp16"
    if half:
        vel, col = vel.astype(np.float16).astype(f32), col.astype(np.float16).astype(f32)
    got = {}
    for lds in ("1", "inline", "0"):
        knob("ADVECT_LDS", "0" if lds == "0" else "2")      # 2 = the LDS path also below the size where it pays
        knob("ADVECT_DEFER", "0" if lds == "inline" else "1")   # far-tracing voxels: k_advect_far | gathers inside the kernel
        f = make(dims, advect_address=address, storage=storage)
        dt = f32(f.default_time_step())
        f.upload(fx.FIELD_VELOCITY, vel); f.upload(fx.FIELD_COLOR, col)
        f.UpdateFrame(dt, 0)
        f.Advect()
        f.Synchronize()
        got[lds] = (f.download(fx.FIELD_VELOCITY1), f.download(fx.FIELD_COLOR))
    for k in ("1", "inline"):
        assert np.array_equal(got[k][0].view(np.uint32), got["0"][0].view(np.uint32)), k
        assert np.array_equal(got[k][1].view(np.uint32), got["0"][1].view(np.uint32)), k
    vo, co = orc.advect(vel, col, dt, address=int(address == "mirror"), half=half)
    assert rel_l2(got["1"][0], vo) < (1e-4 if half else 1e-6) and rel_l2(got["1"][1], co) < (1e-4 if half else 1e-6)
    z, y, x = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij", sparse=True)
    far = ((x + .5) / X - .5) ** 2 + ((y + .5) / Y - .1) ** 2 + ((z + .5) / Z - .5) ** 2 > (1.5 / 16) ** 2
    assert np.array_equal(got["1"][0][:, far], vo[:, far]) and np.array_equal(got["1"][1][far], co[far])


def test_advect_deferred_voxels_over_changing_flows(knob):
    """k_advect_lds puts the voxels that trace beyond its staged window on a list, k_advect_far advects them afterwards; the list's two
    totals alternate launch by launch.  Five advections of one context whose flow swings between "every voxel far" and "none", with a
    launch that does not defer in between: each equals the gather kernel's result bit for bit"""
    dims = (128, 128, 48)
    X, Y, Z = dims
    rng = np.random.default_rng(5)
    col = rng.random((Z, Y, X, 4)).astype(f32)
    knob("ADVECT_LDS", "2")
    f = make(dims)
    knob("ADVECT_LDS", "0")
    ref = make(dims)
    dt = f32(f.default_time_step())
    for k, (scale, defer) in enumerate([(12.0, "1"), (0.2, "1"), (3.0, "0"), (12.0, "1"), (1.0, "1"), (0.0, "1")]):
        vel = (rng.standard_normal((3, Z, Y, X)) * scale).astype(f32)
        out = []
        for g, lds in ((f, "2"), (ref, "0")):
            knob("ADVECT_LDS", lds)
            knob("ADVECT_DEFER", defer)
            g.upload(fx.FIELD_VELOCITY, vel); g.upload(fx.FIELD_COLOR, col)
            g.UpdateFrame(dt, 0)
            g.Advect()
            out.append((g.download(fx.FIELD_VELOCITY1), g.download(fx.FIELD_COLOR)))
        assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32)), k
        assert np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32)), k


@pytest.mark.parametrize("grid,steps,storage", [(256, 40, "fp32"), (128, 60, "fp32"), (256, 24, "fp16"), (150, 30, "fp32"), (150, 24, "fp16")])
def test_round2_kernels_reproduce_the_round1_kernels_over_a_whole_run(grid, steps, storage):
    """the kernels added in round 2 (LDS-staged advection, cooperative three-sweep strips, the X = 128 block kernel, four-cell
    projection / divergence) against the ones they replaced, over a whole run from the zero state at full BASELINE size: the plume
    develops, waves fall off the LDS window, the schedule mixes threes and twos -- and every field ends bit-identical.  (The
    switches are read once per process, hence the child processes.)"""
    import hashlib
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, hashlib, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import fluidx12_amd as fx\n"
        "import os\n"
        "f = fx.Fluid(); assert f.Init(64, 64, (%d, %d, %d), jacobi_iters=40, storage=%r, **({'jacobi_fuse': 1} if os.environ.get('FX_TEST_PLAIN') else {}))\n"
        "dt = np.float32(f.default_time_step())\n"
        "for k in range(%d):\n"
        "    f.UpdateFrame(dt, k %% 3); f.Simulate(k %% 3)\n"
        "f.Synchronize()\n"
        "h = hashlib.sha256()\n"
        "for fid in (fx.FIELD_VELOCITY, fx.FIELD_COLOR, fx.FIELD_PRESSURE):\n"
        "    a = f.download(fid); assert np.isfinite(a).all() and a.any(); h.update(a.tobytes())\n"
        "print('DIGEST', h.hexdigest())\n" % (root, grid, grid, grid, storage, steps))
    old = dict(FLUIDX_ADVECT_LDS="0", FLUIDX_STRIP3_COOP="0", FLUIDX_JACOBI_BLOCK="0", FLUIDX_PROJECT_V4="0")
    # ... and against the plainest kernels the library has: one Jacobi sweep per launch (k_jacobi_v4: none of the register-pair
    # plumbing of fx_pk.h), gather advection, scalar projection (tools/long_run_parity.py runs the same comparison over hundreds of steps)
    plain = dict(old, FX_TEST_PLAIN="1", FLUIDX_ADVECT_FAST="0", FLUIDX_ROW_VW="0", FLUIDX_JACOBI_BLOCKG="0")
    digests = []
    for extra in ({}, old, plain):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        digests.append([l for l in r.stdout.splitlines() if l.startswith("DIGEST")][-1])
    assert digests[0] == digests[1] == digests[2]


@pytest.mark.parametrize("dims,mode,iters", [((512, 512, 1), "faithful", 64), ((512, 512, 1), "fixed", 20), ((100, 100, 1), "faithful", 64),
                                              ((70, 70, 1), "fixed", 13), ((64, 64, 1), "faithful", 9), ((33, 33, 1), "fixed", 40)])
def test_2d_tile_kernel_equals_one_sweep_per_launch(dims, mode, iters):
    """k_jacobi2d_tile (up to eight sweeps per launch on LDS tiles with recomputed halos; the reference's 2-D preset Bin/Fluid2D.bat is
    512 x 512 x 1 with its 64-sweep early-out loop) against one sweep per launch in k_jacobi_generic (jacobi_fuse = 1) over whole
    steps, and its solve alone against the oracle: bit-identical, freeze bytes included"""
    kw = dict(jacobi_mode=mode, jacobi_iters=iters)
    a, b_ = make(dims, **kw), make(dims, jacobi_fuse=1, **kw)
    for k in range(10):
        for f in (a, b_):
            f.UpdateFrame(f32(f.default_time_step()), k % 3)
            f.Simulate(k % 3)
    for fld in (fx.FIELD_PRESSURE, fx.FIELD_VELOCITY, fx.FIELD_COLOR):
        assert np.array_equal(a.download(fld).view(np.uint32), b_.download(fld).view(np.uint32)), fld
    assert np.abs(a.download(fx.FIELD_PRESSURE)).max() > 0
    # the launch count says which kernel ran
    a.timing_enable(True); a.timing_read(True)
    a.UpdateFrame(f32(a.default_time_step()), 1); a.Simulate(1); a.Synchronize()
    t = a.timing_read()
    assert t.jacobi_sweeps == iters and t.jacobi_launches == (iters + 7) // 8
    # the solve alone, from a random state, against the oracle
    X, Y, _ = dims
    rng = np.random.default_rng(3)
    p = (rng.standard_normal((1, Y, X)) * 0.05).astype(f32)
    bb = (rng.standard_normal((1, Y, X)) * 0.02).astype(f32)
    f = make(dims, **kw)
    f.upload(fx.FIELD_PRESSURE, p); f.upload(fx.FIELD_DIVERGENCE, bb)
    f.Jacobi(iters); f.Synchronize()
    want, _ = orc.jacobi(p, bb, iters, mode=int(mode == "faithful"))
    assert np.array_equal(f.download(fx.FIELD_PRESSURE).view(np.uint32), want.view(np.uint32))


def test_2d_tile_kernel_on_more_tiles_than_fit():
    """k_jacobi2d_tile in the reference's mode (freeze bytes) on a 2048 x 2048 grid: 4096 workgroups, far more than are resident at once,
    so late workgroups stage their halos after early ones have stored their cores -- with the freeze bytes updated in place that made a
    halo cell look frozen from level 0; they are double-buffered now.  The solve from a random state against one sweep per
    launch (jacobi_fuse = 1: every cell reads only its own byte), bit for bit, pressure and the number of sweeps that left a cell relaxing."""
    dims, iters = (2048, 2048, 1), 64
    rng = np.random.default_rng(5)
    p = (rng.standard_normal((1, 2048, 2048)) * 0.05).astype(f32)
    bb = (rng.standard_normal((1, 2048, 2048)) * 0.02).astype(f32)
    out = []
    for fuse in (0, 1):
        f = make(dims, jacobi_mode="faithful", jacobi_iters=iters, jacobi_fuse=fuse)
        f.upload(fx.FIELD_PRESSURE, p); f.upload(fx.FIELD_DIVERGENCE, bb)
        f.timing_enable(True); f.timing_read(True)
        for _ in range(3):                       # three solves in a row: the mask is cleared in between and the buffers have swapped an odd number of times
            f.Jacobi(iters)
        f.Synchronize()
        t = f.timing_read()
        assert t.jacobi_launches == (3 * 8 if fuse == 0 else 3 * iters)
        out.append(f.download(fx.FIELD_PRESSURE).view(np.uint32))
        f.Release()
    assert np.array_equal(out[0], out[1])
