"""z-slab decomposition on ONE GPU through the loop-back transport (fx_comm_init_local): several slab
contexts in one process exchange halo planes by device-to-device copies with exactly the halo geometry
and phase schedule the RCCL transport uses.  The decomposed run must equal the single-domain run
bit-for-bit (same arithmetic per cell), which in turn is parity-checked against the oracle elsewhere."""
import numpy as np
import pytest

import fluidx12_amd as fx
from fluidx12_amd import capi
from oracle import orc          # checker only

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(autouse=True, params=["shared", "peer"])
def group_kind(request):
    """every test of this module runs twice: on the shared-stream in-process group (fx_comm_init_local: all members on one stream) and
    on the peer group (fx_comm_init_peer: every member on its own compute / comm / face streams, ordered by events only -- the members
    really run concurrently, and an ordering bug shows as a wrong bit)"""
    from fluidx12_amd import fluid as fluid_mod
    old = fluid_mod.default_local_group
    fluid_mod.default_local_group = request.param
    yield request.param
    fluid_mod.default_local_group = old


def run_single(dims, steps, **kw):
    f = fx.Fluid()
    assert f.Init(800, 800, dims, **kw)
    for k in range(steps):
        f.UpdateFrame(f32(f.default_time_step()), k % 3)
        f.Simulate(k % 3)
    f.Synchronize()
    return f


def run_slabs(dims, steps, nranks, **kw):
    X, Y, Z = dims
    fl = []
    for r in range(nranks):
        z0, z1 = r * Z // nranks, (r + 1) * Z // nranks
        f = fx.Fluid()
        assert f.Init(800, 800, dims, slab=(z0, z1 - z0), **kw), f.last_status
        fl.append(f)
    fx.comm_init_local(fl)
    for k in range(steps):
        fl[0].UpdateFrame(f32(fl[0].default_time_step()), k % 3)      # rank 0 drives the loop-back group
        fl[0].Simulate(k % 3)
    fl[0].Synchronize()
    return fl


def gather(fl, field, axis):
    return np.concatenate([f.download(field) for f in fl], axis=axis)


@pytest.mark.parametrize("overlap", [3, 2, 1, 0])
@pytest.mark.parametrize("nranks", [2, 4])
@pytest.mark.parametrize("iters,hj", [(40, 4), (10, 3), (7, 8), (5, 1)])
def test_slabs_equal_single_domain(nranks, iters, hj, overlap):
    """overlap=True: exchanges on the side stream behind interior work (face planes first); False: everything on one stream"""
    dims = (64, 64, 64)
    steps = 6
    ref = run_single(dims, steps, jacobi_iters=iters)
    fl = run_slabs(dims, steps, nranks, jacobi_iters=iters, halo_jacobi=hj, halo_advect=6, overlap=overlap)
    assert np.array_equal(gather(fl, fx.FIELD_VELOCITY, 1), ref.download(fx.FIELD_VELOCITY))
    assert np.array_equal(gather(fl, fx.FIELD_COLOR, 0), ref.download(fx.FIELD_COLOR))
    assert np.array_equal(gather(fl, fx.FIELD_PRESSURE, 0), ref.download(fx.FIELD_PRESSURE))


@pytest.mark.parametrize("storage", ["fp32", "fp16"])
def test_field_digest_is_independent_of_the_decomposition_and_sees_one_bit(storage):
    """fx_field_digest (what `bench.py --gpus N` certifies its timed steps with): the digest of a range of planes is the same whether
    one context owns them or three slabs do (each digests the part it owns: the words add), differs between fields, ranges and
    storages, refuses planes a context does not own, and changes when ONE bit of one element changes"""
    dims = (64, 64, 48)
    ref = run_single(dims, 5, jacobi_iters=8, storage=storage)
    fl = run_slabs(dims, 5, 3, jacobi_iters=8, halo_jacobi=4, halo_advect=6, storage=storage)
    M = (1 << 128) - 1

    def add(a, b):                                   # the two 64-bit words wrap separately
        return (((a >> 64) + (b >> 64)) & ((1 << 64) - 1)) << 64 | ((a + b) & ((1 << 64) - 1))
    seen = set()
    for field in (fx.FIELD_VELOCITY, fx.FIELD_VELOCITY1, fx.FIELD_COLOR, fx.FIELD_PRESSURE, fx.FIELD_DIVERGENCE):
        whole = ref.digest(field)
        assert 0 < whole <= M and whole not in seen
        seen.add(whole)
        parts = 0
        for r, f in enumerate(fl):
            z0, nz = r * 16, 16
            d = f.digest(field)                                          # all owned planes
            assert d == f.digest(field, z0, nz) == ref.digest(field, z0, nz)
            parts = add(parts, d)
        assert parts == whole
        assert ref.digest(field, 3, 7) != ref.digest(field, 3, 8)
    with pytest.raises(fx.FluidxError):
        fl[1].digest(fx.FIELD_PRESSURE, 0, 16)                          # planes of rank 0
    p = ref.download(fx.FIELD_PRESSURE)
    before = ref.digest(fx.FIELD_PRESSURE)
    p.view(np.uint32)[20, 31, 17] ^= np.uint32(1)
    ref.upload(fx.FIELD_PRESSURE, p)
    after = ref.digest(fx.FIELD_PRESSURE)
    assert after != before
    assert ref.digest(fx.FIELD_PRESSURE, 0, 16) == fl[0].digest(fx.FIELD_PRESSURE)          # the planes below the flipped bit: untouched
    assert ref.digest(fx.FIELD_PRESSURE, 21, 27) == add(fl[1].digest(fx.FIELD_PRESSURE, 21, 11), fl[2].digest(fx.FIELD_PRESSURE))


@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("fuse,iters", [(2, 12), (4, 12), (2, 11), (3, 13)])
def test_slabs_with_temporal_blocking(fuse, iters, overlap):
    dims = (64, 64, 96)
    ref = run_single(dims, 4, jacobi_iters=iters, jacobi_fuse=1)
    fl = run_slabs(dims, 4, 3, jacobi_iters=iters, halo_jacobi=4, halo_advect=8, jacobi_fuse=fuse, overlap=overlap)
    assert np.array_equal(gather(fl, fx.FIELD_PRESSURE, 0), ref.download(fx.FIELD_PRESSURE))
    assert np.array_equal(gather(fl, fx.FIELD_VELOCITY, 1), ref.download(fx.FIELD_VELOCITY))


@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("dims,nranks,hj", [((512, 512, 48), 3, 8), ((512, 512, 32), 2, 6), ((256, 256, 64), 2, 8)])
def test_slabs_with_register_strips(dims, nranks, hj, overlap):
    """the two-sweep register-strip Jacobi kernels (X = 512: the wide two-float4-per-lane kernel; X = 256: the
    float4-per-lane one) inside the slab schedule: shrinking ranges, odd remainders and slab boundaries equal the
    one-sweep-per-launch single-domain run bit-for-bit"""
    ref = run_single(dims, 3, jacobi_iters=14, jacobi_fuse=1)
    fl = run_slabs(dims, 3, nranks, jacobi_iters=14, halo_jacobi=hj, halo_advect=8, jacobi_fuse=2, overlap=overlap)
    assert np.array_equal(gather(fl, fx.FIELD_PRESSURE, 0), ref.download(fx.FIELD_PRESSURE))
    assert np.array_equal(gather(fl, fx.FIELD_VELOCITY, 1), ref.download(fx.FIELD_VELOCITY))
    assert np.array_equal(gather(fl, fx.FIELD_COLOR, 0), ref.download(fx.FIELD_COLOR))


@pytest.mark.parametrize("overlap", [True, False])
def test_slabs_x128_block_kernel(overlap):
    """X = 128 slabs: the rounds run in k_jacobi_block2 on the shrinking plane ranges (ragged depths, blocks cut by the range end)"""
    dims = (128, 128, 72)
    ref = run_single(dims, 3, jacobi_iters=13, jacobi_fuse=1)
    fl = run_slabs(dims, 3, 3, jacobi_iters=13, halo_jacobi=5, halo_advect=6, overlap=overlap)
    assert np.array_equal(gather(fl, fx.FIELD_PRESSURE, 0), ref.download(fx.FIELD_PRESSURE))
    assert np.array_equal(gather(fl, fx.FIELD_VELOCITY, 1), ref.download(fx.FIELD_VELOCITY))


@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("dims,nranks", [((150, 150, 60), 3), ((100, 100, 33), 2)])
def test_slabs_general_block_kernel(dims, nranks, overlap):
    """rows that fit no strip kernel (150 wide: the reference's GI preset) in slabs: the rounds run two sweeps per launch in
    k_jacobi_blockg on the shrinking plane ranges; divergence / projection in the three-cells-per-thread kernels"""
    ref = run_single(dims, 3, jacobi_iters=13, jacobi_fuse=1)
    fl = run_slabs(dims, 3, nranks, jacobi_iters=13, halo_jacobi=5, halo_advect=6, overlap=overlap)
    fl[0].timing_enable(True)
    fl[0].UpdateFrame(f32(fl[0].default_time_step()), 0)
    fl[0].Simulate(0)
    fl[0].Synchronize()
    t = fl[0].timing_read()
    assert t.jacobi_sweeps >= 13 and t.jacobi_launches < t.jacobi_sweeps      # fused launches took part
    ref.UpdateFrame(f32(ref.default_time_step()), 0)
    ref.Simulate(0)
    ref.Synchronize()
    assert np.array_equal(gather(fl, fx.FIELD_PRESSURE, 0), ref.download(fx.FIELD_PRESSURE))
    assert np.array_equal(gather(fl, fx.FIELD_VELOCITY, 1), ref.download(fx.FIELD_VELOCITY))


@pytest.mark.parametrize("overlap", [True, False])
def test_slabs_faithful_mode(overlap):
    """reference-faithful Jacobi (64-sweep cap, per-cell freeze at |delta| < 1e-3): the freeze mask of the face planes
    travels with the pressure halo, so the decomposed run freezes exactly the cells the single domain freezes"""
    dims = (48, 48, 48)
    ref = run_single(dims, 5, jacobi_mode="faithful", jacobi_iters=64)
    fl = run_slabs(dims, 5, 3, jacobi_mode="faithful", jacobi_iters=64, halo_jacobi=2, halo_advect=6, overlap=overlap)
    assert np.array_equal(gather(fl, fx.FIELD_PRESSURE, 0), ref.download(fx.FIELD_PRESSURE))
    assert np.array_equal(gather(fl, fx.FIELD_VELOCITY, 1), ref.download(fx.FIELD_VELOCITY))


@pytest.mark.parametrize("dims,nranks,hj,ha,storage,address,steps", [((64, 64, 96), 3, 8, 6, "fp16", "clamp", 8), ((48, 48, 64), 2, 4, 10, "fp32", "mirror", 6),
                                                                     ((150, 150, 60), 3, 5, 8, "fp16", "clamp", 5), ((128, 128, 100), 4, 8, 8, "fp32", "clamp", 6),
                                                                     ((64, 64, 45), 2, 6, 6, "fp32", "clamp", 9)])
def test_slabs_run_the_reference_s_solve_on_the_sparse_solver(dims, nranks, hj, ha, storage, address, steps):
    """the reference's own configuration (<= 64 sweeps, per-cell early-out) on slab ranks takes the sparse solver of fx_jacobi_freeze.hip
    (a view of each rank's planes: dense sweep over the owned ones, tile cones reaching into the halo, four planes of pressure + mask
    behind every launch) -- every field and the number of executed sweeps equal the single domain's, bit for bit; uneven slabs, halos that
    are no multiple of the tile depth, rows of 150 cells, both storages and samplers"""
    kw = dict(jacobi_mode="faithful", jacobi_iters=64, storage=storage, advect_address=address)
    ref = run_single(dims, steps, **kw)
    fl = run_slabs(dims, steps, nranks, halo_jacobi=hj, halo_advect=ha, **kw)
    for f in [ref] + fl:
        f.timing_read(True)
    for f in (ref, fl[0]):                                   # one more step, counted: the fast path must be the one that runs
        f.UpdateFrame(f32(f.default_time_step()), 0)
        f.Simulate(0)
    ref.Synchronize(); fl[0].Synchronize()
    t = ref.timing_read()
    assert t.freeze_solves == 1 and t.freeze_sweeps >= 2
    most = 0
    for f in fl:
        tt = f.timing_read()
        assert tt.freeze_solves == 1                          # (a slab's own count of sweeps covers its planes only: <= the domain's)
        assert tt.freeze_sweeps <= t.freeze_sweeps
        most = max(most, tt.freeze_sweeps)
    assert most == t.freeze_sweeps                            # the chain's count (the last level that left a cell relaxing anywhere) = the domain's
    assert np.array_equal(gather(fl, fx.FIELD_PRESSURE, 0), ref.download(fx.FIELD_PRESSURE))
    assert np.array_equal(gather(fl, fx.FIELD_VELOCITY, 1), ref.download(fx.FIELD_VELOCITY))
    assert np.array_equal(gather(fl, fx.FIELD_COLOR, 0), ref.download(fx.FIELD_COLOR))


def test_uneven_slabs_and_fp16():
    dims = (48, 48, 40)
    ref = run_single(dims, 4, jacobi_iters=12, storage="fp16")
    fl = run_slabs(dims, 4, 3, jacobi_iters=12, storage="fp16", halo_jacobi=4, halo_advect=8)
    assert np.array_equal(gather(fl, fx.FIELD_VELOCITY, 1), ref.download(fx.FIELD_VELOCITY))
    assert np.array_equal(gather(fl, fx.FIELD_COLOR, 0), ref.download(fx.FIELD_COLOR))


@pytest.mark.parametrize("overlap", [1, 2])
@pytest.mark.parametrize("dims", [(256, 256, 400), (512, 512, 260)])
def test_thick_slabs_take_the_three_sweep_kernel(dims, overlap):
    """thick slabs run their serial rounds as 4 + 4 sweeps (X = 256: k_jacobi_strip4o from 1.5 M cells, X = 512: k_jacobi_strip4x from 16.8 M)
    on the shrinking trapezoid ranges, halo planes included: bit-identical to one sweep per launch on the single domain"""
    ref = run_single(dims, 2, jacobi_iters=19, jacobi_fuse=1)
    fl = run_slabs(dims, 2, 2, jacobi_iters=19, halo_jacobi=8, halo_advect=8, overlap=overlap)
    fl[0].timing_enable(True)
    fl[0].UpdateFrame(f32(fl[0].default_time_step()), 2)
    fl[0].Simulate(2)
    fl[0].Synchronize()
    t = fl[0].timing_read()
    if overlap == 1:
        assert t.jacobi_sweeps == 19 and t.jacobi_launches == 5 and t.jacobi_main_sweeps == 19  # 4+4, 4+4, 3 (no launch of a round is shorter than the one before it)
    # overlap 2: the interior of every round runs as 2 + 3 + 3 beside the single-sweep face chains
    ref.UpdateFrame(f32(ref.default_time_step()), 2)
    ref.Simulate(2)
    ref.Synchronize()
    assert np.array_equal(gather(fl, fx.FIELD_PRESSURE, 0), ref.download(fx.FIELD_PRESSURE))
    assert np.array_equal(gather(fl, fx.FIELD_VELOCITY, 1), ref.download(fx.FIELD_VELOCITY))


@pytest.mark.parametrize("dims,slabs,hj,ha,iters,rnd", [((320, 320, 20), [(0, 12), (12, 8)], 6, 7, 11, 4), ((256, 256, 14), [(0, 9), (9, 5)], 4, 5, 8, 4),
                                                    ((512, 512, 13), [(0, 8), (8, 5)], 5, 5, 10, 5), ((264, 264, 26), [(0, 17), (17, 9)], 7, 7, 19, 4)])
def test_neighbours_that_take_different_launches_for_the_same_sweeps(dims, slabs, hj, ha, iters, rnd):
    """the serial schedule lets every member of a group compose a round from the launches ITS slab prefers: 12 planes of 320 x 320 run
    fours where the 8-plane neighbour runs single sweeps, and the two end a round in different pressure buffers.  The pressure exchange
    takes every member's own current buffer (it took the lead's index for all of them: found by the wide-row slab fuzz of round 6 --
    in-process groups only, a rank of an RCCL chain is its own lead); bit-identical to the single domain"""
    ref = fx.Fluid()
    assert ref.Init(800, 800, dims, jacobi_iters=iters, jacobi_fuse=1)
    fl = []
    for z0, nz in slabs:
        f = fx.Fluid()
        assert f.Init(800, 800, dims, slab=(z0, nz), halo_advect=ha, halo_jacobi=hj, jacobi_iters=iters), f.last_status
        fl.append(f)
    fx.comm_init_local(fl)
    from fluidx12_amd import capi
    for f in fl:
        f.set_option(capi.OPT_OVERLAP, 0)
        f.set_option(capi.OPT_JACOBI_ROUND, rnd)
    fl[0].timing_enable(True); fl[1].timing_enable(True)
    dt = f32(ref.default_time_step())
    for k in range(3):
        ref.UpdateFrame(dt, k % 3); ref.Simulate(k % 3)
        fl[0].UpdateFrame(dt, k % 3); fl[0].Simulate(k % 3)
    ref.Synchronize(); fl[0].Synchronize()
    launches = [f.timing_read().jacobi_launches for f in fl]
    assert launches[0] != launches[1], launches                      # the case this test is about
    assert np.array_equal(gather(fl, fx.FIELD_PRESSURE, 0), ref.download(fx.FIELD_PRESSURE))
    assert np.array_equal(gather(fl, fx.FIELD_VELOCITY, 1), ref.download(fx.FIELD_VELOCITY))
    assert np.abs(ref.download(fx.FIELD_PRESSURE)).max() > 0


def test_switching_to_the_overlapped_schedule_after_the_members_buffers_diverged():
    """... and when a step ends with the members' pressures in different buffers (12 sweeps in rounds of 4: three launches of four on the
    12-plane slab, twelve single sweeps on its 8-plane neighbour), the overlapped schedule -- an option at run time -- starts every
    member's round from ITS buffer (it took the lead's for all)"""
    dims, slabs = (320, 320, 20), [(0, 12), (12, 8)]
    ref = fx.Fluid()
    assert ref.Init(800, 800, dims, jacobi_iters=12, jacobi_fuse=1)
    fl = []
    for z0, nz in slabs:
        f = fx.Fluid()
        assert f.Init(800, 800, dims, slab=(z0, nz), halo_advect=7, halo_jacobi=6, jacobi_iters=12), f.last_status
        fl.append(f)
    fx.comm_init_local(fl)
    from fluidx12_amd import capi
    dt = f32(ref.default_time_step())
    k = 0
    for overlap, rnd, steps in ((0, 4, 1), (2, 2, 2), (0, 4, 1), (3, 2, 2)):
        for f in fl:
            f.set_option(capi.OPT_OVERLAP, overlap)
            f.set_option(capi.OPT_JACOBI_ROUND, rnd)
        for _ in range(steps):
            ref.UpdateFrame(dt, k % 3); ref.Simulate(k % 3)
            fl[0].UpdateFrame(dt, k % 3); fl[0].Simulate(k % 3)
            k += 1
        ref.Synchronize(); fl[0].Synchronize()
        assert np.array_equal(gather(fl, fx.FIELD_PRESSURE, 0), ref.download(fx.FIELD_PRESSURE)), (overlap, rnd)
    assert np.array_equal(gather(fl, fx.FIELD_VELOCITY, 1), ref.download(fx.FIELD_VELOCITY))
    assert np.abs(ref.download(fx.FIELD_PRESSURE)).max() > 0


def test_halo_overflow_is_reported():
    """a back-trace that leaves the exchanged halo must be reported, not silently clamped"""
    dims = (32, 32, 32)
    fl = []
    for r in range(2):
        f = fx.Fluid()
        assert f.Init(800, 800, dims, slab=(r * 16, 16), halo_advect=1, halo_jacobi=1, jacobi_iters=4)
        fl.append(f)
    fx.comm_init_local(fl)
    vel = np.zeros((3, 16, 32, 32), f32)
    vel[2] = 3.0                                     # 3 * dt * Z = 6 cells of z reach > 1-plane halo
    for f in fl:
        f.upload(fx.FIELD_VELOCITY, vel)
    fl[0].UpdateFrame(f32(2.0 / 32), 0)
    fl[0].Simulate(0)
    with pytest.raises(fx.FluidxError) as e:
        fl[0].Synchronize()
    assert e.value.status == capi.FX_E_HALO


def test_halo_overflow_counts_exchanged_planes_not_allocated_ones():
    """the allocation is max(halo_advect, halo_jacobi) planes wide but only halo_advect planes are refreshed before the
    advection: a 3-plane reach with halo_advect = 2, halo_jacobi = 8 reads stale planes and must be reported"""
    dims = (32, 32, 32)
    fl = []
    for r in range(2):
        f = fx.Fluid()
        assert f.Init(800, 800, dims, slab=(r * 16, 16), halo_advect=2, halo_jacobi=8, jacobi_iters=8)
        fl.append(f)
    fx.comm_init_local(fl)
    vel = np.zeros((3, 16, 32, 32), f32)
    vel[2] = 1.5                                     # 1.5 * dt * Z = 3 cells of z reach: inside the allocation, outside the exchange
    for f in fl:
        f.upload(fx.FIELD_VELOCITY, vel)
    fl[0].UpdateFrame(f32(2.0 / 32), 0)
    fl[0].Simulate(0)
    with pytest.raises(fx.FluidxError) as e:
        fl[0].Synchronize()
    assert e.value.status == capi.FX_E_HALO


def test_rccl_transport_single_rank():
    """the RCCL transport on the one GPU we have: librccl is dlopen()ed, a unique id is created, ncclCommInitRank
    succeeds for a 1-rank world and a step runs through the same phase code (no neighbour => no send/recv)"""
    uid = fx.comm_unique_id()
    assert len(uid) >= 128
    dims = (32, 32, 32)
    ref = run_single(dims, 2, jacobi_iters=8)
    f = fx.Fluid()
    assert f.Init(800, 800, dims, jacobi_iters=8)
    f.comm_init_rank(uid, 0, 1)
    for k in range(2):
        f.UpdateFrame(f32(f.default_time_step()), k)
        f.Simulate(k)
    f.Synchronize()
    assert np.array_equal(f.download(fx.FIELD_VELOCITY), ref.download(fx.FIELD_VELOCITY))


def test_schedule_options_at_run_time():
    """fx_set_option switches the schedule between steps; every setting continues the same bit-exact trajectory"""
    dims = (64, 64, 96)
    ref = run_single(dims, 6, jacobi_iters=16)
    fl = []
    for r in range(3):
        f = fx.Fluid()
        assert f.Init(800, 800, dims, slab=(r * 32, 32), jacobi_iters=16, halo_jacobi=8, halo_advect=6)
        fl.append(f)
    fx.comm_init_local(fl)
    settings = [(2, 8), (1, 8), (2, 4), (0, 3), (2, 2), (2, 1)]
    for k, (ov, rnd) in enumerate(settings):
        for f in fl:
            f.set_option(capi.OPT_OVERLAP, ov)
            f.set_option(capi.OPT_JACOBI_ROUND, rnd)
        fl[0].UpdateFrame(f32(fl[0].default_time_step()), k % 3)
        fl[0].Simulate(k % 3)
    fl[0].Synchronize()
    assert np.array_equal(gather(fl, fx.FIELD_VELOCITY, 1), ref.download(fx.FIELD_VELOCITY))
    assert np.array_equal(gather(fl, fx.FIELD_PRESSURE, 0), ref.download(fx.FIELD_PRESSURE))
    with pytest.raises(fx.FluidxError):
        fl[0].set_option(capi.OPT_JACOBI_ROUND, 9)


def test_multi_gpu_render_by_colour_gather():
    """row f-3, the exact way: the slabs' colour planes are gathered into a render-only whole-grid context, which then
    renders the very picture of the single-domain run (light map, cube map and resolved target bit-identical)"""
    dims, vp, steps = (48, 48, 48), (320, 240), 8
    view, proj, eye = fx.default_camera(*vp)

    def render(f):
        f.UpdateFrame(0.0, 0, view, proj, eye)
        f.ClearRenderTarget()
        f.Render(0, fx.Fluid.OPTIMIZED, to_target=True)
        f.Synchronize()
        return f.download(fx.FIELD_LIGHTMAP), f.download(fx.FIELD_CUBEMAP), f.download(fx.FIELD_TARGET)

    ref = fx.Fluid()
    assert ref.Init(vp[0], vp[1], dims, jacobi_iters=16)
    for k in range(steps):
        ref.UpdateFrame(f32(ref.default_time_step()), k % 3)
        ref.Simulate(k % 3)
    want = render(ref)
    assert want[1][..., 3].max() > 20

    fl = run_slabs(dims, steps, 3, jacobi_iters=16, halo_jacobi=4, halo_advect=6)
    full = fx.Fluid()
    assert full.Init(vp[0], vp[1], dims, render_only=True)
    fl[0].gather_color(full, root=1)
    got = render(full)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    assert np.array_equal(full.download(fx.FIELD_COLOR), ref.download(fx.FIELD_COLOR))
    # ... and that picture is the ORACLE's: the gathered colour field marched, lit and resolved on the CPU (light map up to rare
    # R11G11B10 rounding flips, cube map <= 1 LSB, the resolve of the gathered context's own cube map bit for bit)
    col = full.download(fx.FIELD_COLOR)
    fr, lod, rs, mask, _ = orc.update_frame(view, proj, eye, vp[0], vp[1], dims[0], 192)
    fi = full.frame_info()
    assert (fi.cube_lod, fi.ray_samples, fi.visibility_mask) == (lod, rs, mask)
    lm_ref = orc.raymarch_light(col, fr, 64, False, 2)
    assert (got[0] != lm_ref).mean() < 1e-3 and np.abs(got[0] - lm_ref).max() <= np.abs(lm_ref).max() * 2.0 ** -5
    _, cu = orc.raymarch_view(col, lm_ref, fr, dims[0] >> lod, mask, rs, 64, False, True)
    dcube = np.abs(got[1].astype(np.int32) - cu.astype(np.int32))
    assert dcube.max() <= 1 and (dcube > 0).mean() <= 0.02
    wvp_i = np.array(list(fi.world_view_proj_i), f32).reshape(4, 4)
    out, cov = orc.resolve_cube(got[1], fr, wvp_i, vp[0], vp[1])
    target = np.empty((vp[1], vp[0], 4), np.uint8)
    target[...] = (51, 51, 51, 0)
    assert np.array_equal(got[2], orc.blend_premultiplied(out, cov, target)) and cov.mean() > 0.01
    # a render-only context holds no simulation state
    with pytest.raises(fx.FluidxError):
        full.Simulate(0)
    with pytest.raises(fx.FluidxError):
        full.download(fx.FIELD_PRESSURE)
    # mismatching target
    other = fx.Fluid()
    assert other.Init(vp[0], vp[1], (48, 48, 24), render_only=True)
    with pytest.raises(fx.FluidxError):
        fl[0].gather_color(other)
    assert fx.Fluid().Init(64, 64, (32, 32, 32), slab=(0, 16), render_only=True) is False


def test_rccl_gather_single_rank():
    """the RCCL transport's gather on the one GPU we have (1-rank communicator: the root's own part is a device copy)"""
    dims = (32, 32, 32)
    f = fx.Fluid()
    assert f.Init(200, 150, dims, jacobi_iters=8)
    f.comm_init_rank(fx.comm_unique_id(), 0, 1)
    for k in range(4):
        f.UpdateFrame(f32(f.default_time_step()), k % 3)
        f.Simulate(k % 3)
    full = fx.Fluid()
    assert full.Init(200, 150, dims, render_only=True)
    f.gather_color(full, root=0, slabs=[(0, 32)])
    f.Synchronize()
    full.Synchronize()
    assert np.array_equal(full.download(fx.FIELD_COLOR), f.download(fx.FIELD_COLOR))


def test_loopback_group_survives_any_destruction_order():
    """the members of a loop-back group share one compute stream owned by the group: destroying the driver first must
    neither crash nor leave the others usable by accident"""
    dims = (32, 32, 32)
    fl = run_slabs(dims, 2, 2, jacobi_iters=4, halo_jacobi=2, halo_advect=4)
    fl[0].Release()                                      # the driver goes first
    with pytest.raises(fx.FluidxError):
        fl[1].UpdateFrame(f32(0.01), 0)
    with pytest.raises(fx.FluidxError):
        fl[1].Simulate(0)
    assert fl[1].download(fx.FIELD_PRESSURE).shape == (16, 32, 32)    # read-back of what it holds still works
    fl[1].Release()
    fl = run_slabs(dims, 2, 3, jacobi_iters=4, halo_jacobi=2, halo_advect=4)
    fl[2].Release(); fl[1].Release()
    with pytest.raises(fx.FluidxError):
        fl[0].Simulate(0)
    fl[0].Release()


def test_slab_descriptor_validation():
    f = fx.Fluid()
    assert f.Init(800, 800, (32, 32, 32), slab=(0, 4), halo_advect=8) is False     # halo wider than the slab
    assert f.Init(800, 800, (32, 32, 32), slab=(16, 32)) is False                  # slab leaves the grid


def test_bench_multi_rank_path_in_loopback():
    """bench.py's N-rank code path (grid choice, slabs, halos, schedule timing, JSON) on one GPU via --loopback"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--loopback", "2", "--grid", "64", "--iters", "8",
                          "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and "LOOP-BACK" in d["data"] and d["config"]["grid"] == [64, 64, 128]
    sched = d["config"]["schedule"]
    assert len(sched["candidates"]) == 5 and (sched["overlap"], sched["jacobi_round"]) in [(1, 8), (2, 8), (2, 4), (0, 8), (0, 4)]
    assert d["value"] > 0 and d["scaling"] == "weak"
    # the line certifies its timed steps: each slab's owned planes == the same planes of ONE domain stepped through the same frames
    assert d["multi_rank_parity"] == "bit-identical" and d["multi_rank_parity_detail"]["steps_replayed"] == 5 * 4 + 1 + 3
    # ... and says so when they are not (fault injection: one bit of slab 1's pressure flips behind the timed steps): exit code 4
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--loopback", "2", "--grid", "64", "--iters", "8", "--steps", "3", "--warmup", "1",
                          "--schedule", "0,8"], capture_output=True, text=True, timeout=300, cwd=root, env=dict(os.environ, FLUIDX_BENCH_FAULT="corrupt:1"))
    assert bad.returncode == 4, (bad.returncode, bad.stderr[-1500:])
    db = json.loads([l for l in bad.stdout.splitlines() if l.startswith("{")][-1])
    assert db["multi_rank_parity"].startswith("rank 1: pressure of planes [64, 128) differs")


@pytest.mark.parametrize("overlap", [3, 2, 0])
def test_slabs_mirror_addressing(overlap):
    """the reference's `Fluid` sampler mode (MIRROR, Fluid.cpp:452) in slabs: mirrored taps only exist at the GLOBAL z faces"""
    dims = (64, 64, 64)
    ref = run_single(dims, 6, jacobi_iters=10, advect_address="mirror", storage="fp16")
    fl = run_slabs(dims, 6, 4, jacobi_iters=10, advect_address="mirror", storage="fp16", halo_jacobi=4, halo_advect=6, overlap=overlap)
    assert np.array_equal(gather(fl, fx.FIELD_VELOCITY, 1), ref.download(fx.FIELD_VELOCITY))
    assert np.array_equal(gather(fl, fx.FIELD_COLOR, 0), ref.download(fx.FIELD_COLOR))
    assert np.array_equal(gather(fl, fx.FIELD_PRESSURE, 0), ref.download(fx.FIELD_PRESSURE))


def test_checkpoint_resume_is_bit_identical_across_decompositions(tmp_path):
    """fx_checkpoint_save / _load: a run resumed from the state file equals the uninterrupted run bit for bit -- also when
    the file was written by a 2-slab chain (both members into the same file) and is read by a 3-slab chain, and with
    fp16 field storage (whose stored halves pass through the file's fp32 exactly)"""
    dims = (64, 64, 72)
    for storage in ("fp32", "fp16"):
        kw = dict(storage=storage, jacobi_iters=12)
        ref = run_single(dims, 9, **kw)                                  # uninterrupted
        path = str(tmp_path / ("state_%s.fxck" % storage))
        fl = run_slabs(dims, 5, 2, halo_jacobi=4, halo_advect=8, **kw)
        for f in fl:
            f.SaveCheckpoint(path)
        hdr = np.fromfile(path, np.uint32, 8)
        assert bytes(hdr[:2].tobytes()) == b"FXCKPT03" and tuple(hdr[2:5]) == dims and hdr[6] == 5
        ck = fx.read_checkpoint(path)                                    # the numpy reader sees what the slabs wrote
        assert ck["grid"] == dims and ck["steps"] == 5
        assert np.array_equal(ck["pressure"], gather(fl, fx.FIELD_PRESSURE, 0)) and np.array_equal(ck["velocity"], gather(fl, fx.FIELD_VELOCITY, 1))
        cells = dims[0] * dims[1] * dims[2]
        assert (tmp_path / ("state_%s.fxck" % storage)).stat().st_size == 64 + 8 * cells * 4 + 8 * dims[2] and ck["complete"].all()
        # (a) single domain resumes
        one = fx.Fluid()
        assert one.Init(800, 800, dims, **kw)
        one.LoadCheckpoint(path)
        # (b) another decomposition resumes
        three = []
        for r in range(3):
            f = fx.Fluid()
            assert f.Init(800, 800, dims, slab=(r * 24, 24), halo_jacobi=3, halo_advect=8, **kw)
            f.LoadCheckpoint(path)
            three.append(f)
        fx.comm_init_local(three)
        for k in range(5, 9):                                            # frame indices continue; any index gives the same arithmetic
            for drv in (one, three[0]):
                drv.UpdateFrame(f32(drv.default_time_step()), k % 3)
                drv.Simulate(k % 3)
        one.Synchronize()
        three[0].Synchronize()
        for field, axis in ((fx.FIELD_VELOCITY, 1), (fx.FIELD_COLOR, 0), (fx.FIELD_PRESSURE, 0)):
            want = ref.download(field)
            assert want.any()
            assert np.array_equal(one.download(field), want), (storage, field)
            assert np.array_equal(gather(three, field, axis), want), (storage, field)
    # a save that only one rank of a chain made: the other rank's planes are not marked complete, nobody resumes from them
    part = str(tmp_path / "partial.fxck")
    fl[0].SaveCheckpoint(part)
    assert not fx.read_checkpoint(part)["complete"][dims[2] // 2:].any() and fx.read_checkpoint(part)["complete"][:dims[2] // 2].all()
    one = fx.Fluid()
    assert one.Init(800, 800, dims, **kw)
    with pytest.raises(fx.FluidxError):
        one.LoadCheckpoint(part)
    # a file of another grid is refused
    other = fx.Fluid()
    assert other.Init(800, 800, (32, 32, 32))
    with pytest.raises(fx.FluidxError):
        other.LoadCheckpoint(path)


def test_early_colour_halo_survives_uploads_pauses_and_level_changes():
    """FX_OPT_OVERLAP 3 sends the colour half of the next advection halo a step early.  Whatever happens between the steps --
    a colour upload (loop-back groups drop the early halo and exchange again), paused frames (the parity does not flip, so the
    early halo is for the wrong buffer), switching the level back and forth, a checkpoint reload -- the run equals the single
    domain bit for bit"""
    dims = (64, 64, 96)
    rng = np.random.default_rng(5)
    ref = fx.Fluid()
    assert ref.Init(800, 800, dims, jacobi_iters=9)
    fl = []
    for r in range(3):
        f = fx.Fluid()
        assert f.Init(800, 800, dims, slab=(r * 32, 32), jacobi_iters=9, halo_jacobi=4, halo_advect=8, overlap=3)
        fl.append(f)
    fx.comm_init_local(fl)
    dt = f32(ref.default_time_step())

    def step(dtv, k):
        for drv in (ref, fl[0]):
            drv.UpdateFrame(dtv, k % 3)
            drv.Simulate(k % 3)

    k = 0
    for _ in range(3):
        step(dt, k); k += 1
    # a colour upload between two steps (same data into both runs)
    col = ref.download(fx.FIELD_COLOR)
    col[40:56] = rng.random(col[40:56].shape).astype(f32) * 0.5
    ref.upload(fx.FIELD_COLOR, col)
    for r, f in enumerate(fl):
        f.upload(fx.FIELD_COLOR, col[r * 32:(r + 1) * 32])
    step(dt, k); k += 1
    step(f32(0.0), k); k += 1                              # paused frame, then running again
    step(dt, k); k += 1
    for lvl in (1, 3, 0, 3, 2, 3):
        for f in fl:
            f.set_option(capi.OPT_OVERLAP, lvl)
        step(dt, k); k += 1
    ref.Synchronize()
    fl[0].Synchronize()
    for field, axis in ((fx.FIELD_VELOCITY, 1), (fx.FIELD_COLOR, 0), (fx.FIELD_PRESSURE, 0)):
        assert np.array_equal(gather(fl, field, axis), ref.download(field)), field


# ---- BASELINE configs[3]: 512^3, 80 sweeps, 8 z-slabs of 64 planes ------------------------------------------------------
_CFG4 = {}


def _config4_reference():
    """the single-domain 512^3 / 80-sweep run from a seeded non-trivial state (computed once per session): every slab schedule
    below must reproduce it bit for bit"""
    if _CFG4:
        return _CFG4
    X = 512
    rng = np.random.default_rng(4096)
    vel = rng.random((3, X, X, X), dtype=f32)
    vel -= f32(0.5)                                       # |u| <= 0.5: z back-trace reach <= 1 cell (+ the impulse), well inside the 8-plane halo
    col = rng.random((X, X, X, 4), dtype=f32)
    p = rng.random((X, X, X), dtype=f32)
    ref = fx.Fluid()
    assert ref.Init(800, 800, (X, X, X), jacobi_iters=80)
    ref.upload(fx.FIELD_VELOCITY, vel); ref.upload(fx.FIELD_COLOR, col); ref.upload(fx.FIELD_PRESSURE, p)   # parity 0: the first advection reads this colour buffer
    for k in range(2):
        ref.UpdateFrame(f32(ref.default_time_step()), k)
        ref.Simulate(k)
    ref.Synchronize()
    _CFG4.update(vel=vel, col=col, p=p, out_v=ref.download(fx.FIELD_VELOCITY), out_c=ref.download(fx.FIELD_COLOR),
                 out_p=ref.download(fx.FIELD_PRESSURE))
    ref.Release()
    return _CFG4


@pytest.mark.parametrize("overlap,rnd", [(1, 8), (2, 8), (3, 8), (2, 4), (0, 8)])
def test_config4_512_cubed_80_sweeps_on_8_slabs(overlap, rnd):
    """BASELINE configs[3] at full size through the loop-back transport: 512^3 fp32, 80 lock-step sweeps, 8 ranks x 64 planes,
    two steps under every slab schedule bench.py can pick == the single 512^3 domain, bit for bit (velocity, colour, pressure)"""
    X, N = 512, 8
    c = _config4_reference()
    fl = []
    for r in range(N):
        f = fx.Fluid()
        assert f.Init(800, 800, (X, X, X), slab=(r * 64, 64), jacobi_iters=80, halo_jacobi=8, halo_advect=8, overlap=overlap), f.last_status
        sl = slice(r * 64, (r + 1) * 64)
        f.upload(fx.FIELD_VELOCITY, c["vel"][:, sl]); f.upload(fx.FIELD_COLOR, c["col"][sl]); f.upload(fx.FIELD_PRESSURE, c["p"][sl])
        fl.append(f)
    fx.comm_init_local(fl)
    for f in fl:
        f.set_option(capi.OPT_JACOBI_ROUND, rnd)
    for k in range(2):
        fl[0].UpdateFrame(f32(fl[0].default_time_step()), k)
        fl[0].Simulate(k)
    fl[0].Synchronize()
    for r, f in enumerate(fl):
        sl = slice(r * 64, (r + 1) * 64)
        assert np.array_equal(f.download(fx.FIELD_PRESSURE), c["out_p"][sl]), r
        assert np.array_equal(f.download(fx.FIELD_VELOCITY), c["out_v"][:, sl]), r
        assert np.array_equal(f.download(fx.FIELD_COLOR), c["out_c"][sl]), r
    for f in fl[::-1]:
        f.Release()


def test_bench_config4_in_loopback():
    """bench.py --config 4 emits BASELINE configs[3] (512^3, 80 sweeps, 8 slabs of 64 planes); here through --loopback 8"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "4", "--loopback", "8", "--steps", "2", "--warmup", "1",
                          "--schedule", "1,8"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["config"]["grid"] == [512, 512, 512] and d["config"]["jacobi_iters"] == 80
    assert "64 planes per rank" in d["config"]["parallelism"] and d["value"] > 0
    assert d["multi_rank_parity"] == "bit-identical"          # eight slabs of 512 x 512 x 64 == the single 512^3 domain


def test_bench_single_gpu_line_is_complete():
    """the N = 1 bench line as the driver runs it (`--steps 20 --warmup 5`, smaller grid here): every figure it promises is there and
    non-zero -- stage times and the roofline launch average from the marked steps (every fourth), the render passes measured behind
    the timed loop (whose last marked step leaves the marks off), the CPU baseline"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--grid", "128", "--steps", "20", "--warmup", "5", "--cpu-budget", "2"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["value"] > 0 and d["ms_per_step"] > 0
    assert d["timing_marks"] == {"every": 4, "marked_steps": 5}
    assert all(d["stage_ms_per_step"][k] > 0 for k in ("advect", "divergence", "jacobi", "project"))
    r = d["roofline"]
    assert r["avg_launch_us"] > 0 and r["achieved"] > 0 and r["launches"] > 0
    # `frac` bounds: the launch's compulsory bytes (p, b read once, p' written once) / its average duration / the 8 TB/s peak, from the same line
    assert 0 < r["frac"] <= 1 and abs(r["frac"] - r["compulsory_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 8e12) < 1e-9
    assert r["frac_algorithmic"] == pytest.approx(r["frac"] * r["sweeps_per_launch"]) and isinstance(r["stale"], bool)
    assert (r["traffic"] is None) == (r["stale"] or r["traffic_source"] is None)
    assert 0 < d["step_compulsory"]["frac_of_peak"] <= 1
    rn = d["render"]
    assert rn["light_pass_ms"] > 0 and rn["view_pass_ms"] > 0 and rn["cube_resolve_ms"] > 0 and rn["direct_march_ms"] > 0 and rn["rays_per_s"] > 0
    # the render leg is pinned to frame 132 whatever --steps / --warmup were, and counts the samples it takes
    assert rn["frame"] == 132 and rn["untimed_steps_to_frame"] == 107
    assert rn["view_samples_taken"] > rn["rays"] and rn["light_samples_taken"] >= 128 ** 3 and rn["samples_per_s"] > 0
    assert 0 < rn["bound"]["light_pass"]["frac_of_hbm_peak"] < 1 and 0 < rn["bound"]["view_pass"]["frac_of_hbm_peak"] < 1
    # ... and the step is timed once more on the developed plume (frames 133-152), beside `value`
    dv = d["developed_plume"]
    assert dv["frames"] == [133, 152] and dv["settling_steps_behind_the_render_leg"] == 40 and dv["ms_per_step"] > 0 and dv["value"] == pytest.approx(128 ** 3 / (dv["ms_per_step"] * 1e-3))
    assert dv["marked_steps"] == 5 and all(dv["stage_ms_per_step"][k] > 0 for k in ("advect", "divergence", "jacobi", "project"))
    # both truths: `value` is the contract's protocol and nothing else (no device wake-up in front of it: `value_cold` says it again),
    # the same steps on a device that is awake are `warm_device`
    assert d["device_preheat"] is None and d["value_cold"] == d["value"]
    assert d["warm_device"]["value"] > 0 and d["warm_device"]["ms_per_step"] > 0
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1


def test_bench_reference_configuration_line():
    """`bench.py --reference-config`: the configuration the reference itself runs (64-sweep cap + early-out, RGBA16F) on the sparse
    solver -- its own roofline object (the dense first sweep) and the executed-sweep statistics"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--reference-config", "--grid", "128", "--steps", "20", "--warmup", "12",
                          "--cpu-budget", "2", "--no-render"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["jacobi_mode"] == "faithful" and d["config"]["jacobi_iters"] == 64 and d["config"]["storage"] == "fp16"
    assert "REFERENCE's own configuration" in d["config"]["workload"] and d["value"] > 0
    r = d["roofline"]
    assert "k_freeze_dense" in r["kernel"] and 0 < r["frac"] <= 1 and r["sweeps_per_launch"] == 1.0
    sp = r["sparse_solver"]
    assert 1 < sp["sweeps_executed_per_solve"] <= 64 and sp["solves"] == 20 and sp["tile_launches_per_step"] == 16
    dv = d["developed_plume"]                                  # (--no-render: the untimed steps to frame 132 are run here)
    assert dv["frames"] == [133, 152] and dv["ms_per_step"] > 0 and 1 < dv["sweeps_executed_per_solve"] <= 64
    assert d["cpu_baseline"]["value"] > 0 and "reference configuration" in d["cpu_baseline"]["sample"]


@pytest.mark.parametrize("overlap", [2, 0])
def test_slabs_with_lds_advection(overlap, knob):
    """k_advect_lds inside the slab schedule (interior range with own-planes-only back-traces, full range after the exchange,
    halo planes as the ring's z-1 / z+1): equal to the single domain advected by k_advect_fast, bit for bit"""
    dims = (64, 64, 96)
    knob("ADVECT_LDS", "0")
    ref = run_single(dims, 6, jacobi_iters=10)
    want = (ref.download(fx.FIELD_VELOCITY), ref.download(fx.FIELD_COLOR), ref.download(fx.FIELD_PRESSURE))
    knob("ADVECT_LDS", "2")
    fl = run_slabs(dims, 6, 2, jacobi_iters=10, halo_jacobi=4, halo_advect=6, overlap=overlap)
    assert np.array_equal(gather(fl, fx.FIELD_VELOCITY, 1), want[0])
    assert np.array_equal(gather(fl, fx.FIELD_COLOR, 0), want[1])
    assert np.array_equal(gather(fl, fx.FIELD_PRESSURE, 0), want[2])


# ---- FX_OPT_ADAPTIVE_HALO: the advection exchange follows the measured need per face -----------------------------------------
def _timed_steps(fl, k0, steps, dt=None):
    for f in fl:
        f.timing_enable(True); f.timing_read(True)
    for k in range(k0, k0 + steps):
        fl[0].UpdateFrame(f32(dt if dt is not None else fl[0].default_time_step()), k % 3)
        fl[0].Simulate(k % 3)
    fl[0].Synchronize()
    return [f.timing_read(True) for f in fl]


@pytest.mark.parametrize("overlap", [2, 0])
def test_adaptive_halo_carries_only_what_the_advection_touches(overlap):
    """with the option on (default) a face carries max(need of its two slabs) planes -- measured behind the projection with the
    advection's own arithmetic -- instead of halo_advect; off, always halo_advect.  Same bits either way (and as the single
    domain); far fewer bytes: the plume of this run never comes near the faces with more than a plane or two of reach"""
    dims, Ha, steps = (64, 64, 96), 10, 6
    ref = run_single(dims, 2 + steps, jacobi_iters=10)
    want = (ref.download(fx.FIELD_VELOCITY), ref.download(fx.FIELD_COLOR), ref.download(fx.FIELD_PRESSURE))
    sent = {}
    for adaptive in (1, 0):
        fl = run_slabs(dims, 2, 3, jacobi_iters=10, halo_jacobi=4, halo_advect=Ha, overlap=overlap)
        for f in fl:
            f.set_option(capi.OPT_ADAPTIVE_HALO, adaptive)
        if not adaptive:                                   # (the record of step 2 is already out: one more step runs on it)
            _timed_steps(fl, 2, 1); steps_left = steps - 1; k0 = 3
        else:
            steps_left, k0 = steps, 2
        t = _timed_steps(fl, k0, steps_left)
        assert np.array_equal(gather(fl, fx.FIELD_VELOCITY, 1), want[0])
        assert np.array_equal(gather(fl, fx.FIELD_COLOR, 0), want[1])
        assert np.array_equal(gather(fl, fx.FIELD_PRESSURE, 0), want[2])
        planes = [x.advect_halo_planes / steps_left for x in t]
        sent[adaptive] = sum(x.exchange_bytes for x in t) / steps_left
        if adaptive:
            assert planes[0] <= 3 and planes[1] <= 6 and planes[2] <= 3, planes      # one face / two faces / one face
        else:
            assert planes == [Ha, 2 * Ha, Ha], planes
    assert sent[1] < 0.6 * sent[0], sent


def test_adaptive_halo_falls_back_when_the_measurement_does_not_cover_the_step():
    """the need was measured with the previous step's dt and on the velocity the projection left: a larger dt, or a velocity
    upload in between, make the next exchange carry halo_advect planes again; every such run equals the single domain"""
    dims, Ha = (64, 64, 128), 8                             # rank 0's only face is plane 32, far from the impulse at plane 64
    rng = np.random.default_rng(9)
    ref = fx.Fluid()
    assert ref.Init(800, 800, dims, jacobi_iters=8)
    fl = run_slabs(dims, 0, 4, jacobi_iters=8, halo_jacobi=4, halo_advect=Ha)
    dt = 0.25 * ref.default_time_step()                     # a calm flow: the doubled step below must not outrun the 8-plane halo

    def both(dtv, k):
        ref.UpdateFrame(f32(dtv), k % 3); ref.Simulate(k % 3)
        return _timed_steps(fl, k, 1, dtv)

    for k in range(3):
        t = both(dt, k)
    assert t[0].advect_halo_planes <= 2                     # measured: a plane or two
    t = both(2 * dt, 3)                                     # a larger step than the measurement covers
    assert t[0].advect_halo_planes == Ha and t[1].advect_halo_planes == 2 * Ha
    t = both(0.5 * dt, 4)                                   # a smaller one is covered
    assert t[0].advect_halo_planes <= 2
    vel = (rng.standard_normal((3,) + dims[::-1]) * 0.3).astype(f32)
    ref.Synchronize()
    ref.upload(fx.FIELD_VELOCITY, vel)
    for r, f in enumerate(fl):
        f.upload(fx.FIELD_VELOCITY, vel[:, r * 32:(r + 1) * 32])
    t = both(0.5 * dt, 5)                                   # the measurement was of another field
    assert t[0].advect_halo_planes == Ha
    t = both(0.5 * dt, 6)
    assert t[0].advect_halo_planes < Ha
    ref.Synchronize()
    for field, axis in ((fx.FIELD_VELOCITY, 1), (fx.FIELD_COLOR, 0), (fx.FIELD_PRESSURE, 0)):
        assert np.array_equal(gather(fl, field, axis), ref.download(field)), field


@pytest.mark.parametrize("overlap", [0, 2])
def test_a_need_beyond_the_allocated_halo_stops_the_step_before_it_touches_a_field(overlap):
    """halo_advect = 1 and a flow that speeds up: at some step the measured need of a face exceeds the allocation.  The NEXT
    fx_simulate then returns FX_E_HALO before anything is enqueued that could read a plane nobody sent: velocity[0] and the
    pressure are still those of the single-domain run.  (Serial schedule: nothing at all was enqueued and no overflow is flagged;
    overlapped: the interior advection, which reads owned planes only, was already on its way and may have flagged its own
    far-tracing voxels -- fx_synchronize then reports the same status once more.)"""
    dims = (32, 32, 32)
    fl = run_slabs(dims, 0, 2, jacobi_iters=8, halo_jacobi=2, halo_advect=1, overlap=overlap)
    ref = fx.Fluid()
    assert ref.Init(800, 800, dims, jacobi_iters=8)
    dt = f32(ref.default_time_step())
    done = None
    for k in range(60):
        fl[0].UpdateFrame(dt, k % 3)
        try:
            fl[0].Simulate(k % 3)
        except fx.FluidxError as e:
            assert e.status == capi.FX_E_HALO
            done = k
            break
        ref.UpdateFrame(dt, k % 3); ref.Simulate(k % 3)
    assert done is not None and done >= 1                  # (the first step has no measurement: it runs on halo_advect planes)
    if overlap == 0:
        fl[0].Synchronize()                                  # no overflow flag: nothing went wrong on the device
    else:
        try:
            fl[0].Synchronize()
        except fx.FluidxError as e:
            assert e.status == capi.FX_E_HALO
    ref.Synchronize()
    assert np.array_equal(gather(fl, fx.FIELD_VELOCITY, 1), ref.download(fx.FIELD_VELOCITY))
    assert np.array_equal(gather(fl, fx.FIELD_PRESSURE, 0), ref.download(fx.FIELD_PRESSURE))


def test_an_overflow_on_one_rank_stops_every_rank_and_blocks_read_back():
    """FX_OPT_ADAPTIVE_HALO off, halo_advect too small for the uploaded velocity: the advection of the step flags the overflow on
    the device; fx_download / fx_checkpoint_save of that context refuse (FX_E_HALO), the next fx_simulate refuses for the whole
    chain, fx_synchronize reports and acknowledges it"""
    dims = (32, 32, 32)
    fl = run_slabs(dims, 0, 2, jacobi_iters=4, halo_jacobi=1, halo_advect=1)
    vel = np.zeros((3, 16, 32, 32), f32)
    vel[2] = 3.0
    for f in fl:
        f.upload(fx.FIELD_VELOCITY, vel)
    fl[0].UpdateFrame(f32(2.0 / 32), 0)
    fl[0].Simulate(0)
    with pytest.raises(fx.FluidxError) as e:
        fl[1].download(fx.FIELD_VELOCITY)
    assert e.value.status == capi.FX_E_HALO
    with pytest.raises(fx.FluidxError) as e:
        fl[1].SaveCheckpoint("/tmp/fluidx_never_written.fxck")
    assert e.value.status == capi.FX_E_HALO
    import os
    assert not os.path.exists("/tmp/fluidx_never_written.fxck")
    fl[0].UpdateFrame(f32(2.0 / 32), 1)
    with pytest.raises(fx.FluidxError) as e:
        fl[0].Simulate(1)
    assert e.value.status == capi.FX_E_HALO
    with pytest.raises(fx.FluidxError) as e:
        fl[0].Synchronize()
    assert e.value.status == capi.FX_E_HALO
    fl[0].Synchronize()                                      # acknowledged
    assert fl[1].download(fx.FIELD_VELOCITY).shape == (3, 16, 32, 32)


def test_the_fault_notice_acknowledges_itself_and_the_chain_runs_on():
    """nobody calls fx_synchronize.  The step after the overflow is refused on every rank (the notice); the
    notice takes the device flag down -- left up, every later step's record would carry it and the chain would alternate between
    one executed and one refused step for ever -- so the two steps after it run.  The faulting ranks still refuse read-back until
    their fx_synchronize has reported the fault."""
    dims = (32, 32, 32)
    fl = run_slabs(dims, 0, 2, jacobi_iters=4, halo_jacobi=1, halo_advect=3)     # (three planes: what the impulse's own flow needs later on)
    vel = np.zeros((3, 16, 32, 32), f32)
    vel[2] = 3.0                                             # traces six cells back
    for f in fl:
        f.set_option(capi.OPT_ADAPTIVE_HALO, 0)              # (the measured need of THIS field would stop step 0 before it overflows)
        f.upload(fx.FIELD_VELOCITY, vel)
    dt = f32(2.0 / 32)
    fl[0].UpdateFrame(dt, 0)
    fl[0].Simulate(0)                                        # overflows on the device
    fl[0].UpdateFrame(dt, 1)
    with pytest.raises(fx.FluidxError) as e:
        fl[0].Simulate(1)                                    # the chain-wide notice
    assert e.value.status == capi.FX_E_HALO
    for f in fl:
        f.upload(fx.FIELD_VELOCITY, np.zeros_like(vel))      # a slow field: nothing overflows any more
    for k in (2, 3, 4):
        fl[0].UpdateFrame(dt, k % 3)
        fl[0].Simulate(k % 3)                                # (before: step 3 was refused again, then 5, 7, ...)
    with pytest.raises(fx.FluidxError) as e:
        fl[1].download(fx.FIELD_VELOCITY)                    # the fault of this rank has not been acknowledged yet
    assert e.value.status == capi.FX_E_HALO
    with pytest.raises(fx.FluidxError) as e:
        fl[0].Synchronize()
    assert e.value.status == capi.FX_E_HALO
    fl[0].Synchronize()
    assert fl[1].download(fx.FIELD_VELOCITY).shape == (3, 16, 32, 32)


def test_overwriting_a_checkpoint_from_one_rank_only_is_refused(tmp_path):
    """the periodic-checkpoint case.  A complete file exists; at a later step only ONE slab saves again (its peer
    crashed, or refused with FX_E_HALO).  The header then names the new save while the peer's planes still carry the old save's
    marks: the load must refuse the mix of two time steps (format 02 accepted it: its marks were 0 / 1)."""
    dims = (32, 32, 32)
    path = str(tmp_path / "periodic.fxck")
    fl = run_slabs(dims, 3, 2, jacobi_iters=6, halo_jacobi=2, halo_advect=6)
    for f in fl:
        f.SaveCheckpoint(path)
    assert fx.read_checkpoint(path)["complete"].all()
    fl2 = run_slabs(dims, 0, 2, jacobi_iters=6, halo_jacobi=2, halo_advect=6)
    for f in fl2:
        f.LoadCheckpoint(path)                               # a complete save loads
    for k in range(2):
        fl[0].UpdateFrame(f32(fl[0].default_time_step()), k)
        fl[0].Simulate(k)
    fl[0].Synchronize()
    fl[0].SaveCheckpoint(path)                               # rank 1 never writes this one
    ck = fx.read_checkpoint(path)
    assert ck["steps"] == 5 and ck["complete"][:16].all() and not ck["complete"][16:].any() and (ck["marks"][16:] == 4).all()
    whole = fx.Fluid()
    assert whole.Init(0, 0, dims, jacobi_iters=6)
    for f in (fl2[1], whole):                                # whoever reads a plane of the older save refuses
        with pytest.raises(fx.FluidxError) as e:
            f.LoadCheckpoint(path)
        assert e.value.status == capi.FX_E_INVALID
    fl[1].SaveCheckpoint(path)                               # the peer catches up: complete again
    for f in fl2 + [whole]:
        f.LoadCheckpoint(path)


def test_fault_acknowledged_before_the_next_step_is_reported_once_more_and_no_further():
    """the usual per-frame order Simulate -> Synchronize -> Simulate: the synchronize reports and acknowledges the fault of the
    rank it happened on; the next fx_simulate still tells EVERY rank, once, and steps nothing; it does not block read-back again, and the
    step after it runs"""
    dims = (32, 32, 32)
    fl = run_slabs(dims, 0, 2, jacobi_iters=4, halo_jacobi=1, halo_advect=1)
    vel = np.zeros((3, 16, 32, 32), f32)
    vel[2] = 3.0
    for f in fl:
        f.upload(fx.FIELD_VELOCITY, vel)
    dt = f32(2.0 / 32)
    fl[0].UpdateFrame(dt, 0)
    fl[0].Simulate(0)                                        # overflows on the device
    with pytest.raises(fx.FluidxError) as e:
        fl[0].Synchronize()
    assert e.value.status == capi.FX_E_HALO
    for f in fl:                                             # acknowledged: read-back works, and a slow field stops the overflow
        assert f.download(fx.FIELD_VELOCITY).shape == (3, 16, 32, 32)
        f.upload(fx.FIELD_VELOCITY, np.zeros_like(vel))
    fl[0].UpdateFrame(dt, 1)
    with pytest.raises(fx.FluidxError) as e:
        fl[0].Simulate(1)                                    # the chain-wide notice of the old fault
    assert e.value.status == capi.FX_E_HALO
    fl[0].Synchronize()                                      # nothing new to report
    assert fl[1].download(fx.FIELD_COLOR).shape == (16, 32, 32, 4)
    fl[0].UpdateFrame(dt, 2)
    fl[0].Simulate(2)
    fl[0].Synchronize()


def test_peer_group_across_two_devices():
    """fx_comm_init_peer with the slabs on different devices: halo planes travel by hipMemcpyPeerAsync out of the neighbour's memory.
    Needs a second GPU (the driver's boxes have one: skipped there)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one device")
    dims, steps = (64, 64, 64), 6
    ref = run_single(dims, steps, jacobi_iters=20)
    fl = []
    for r in range(2):
        f = fx.Fluid()
        assert f.Init(800, 800, dims, slab=(32 * r, 32), jacobi_iters=20, halo_jacobi=4, halo_advect=6, device=r), f.last_status
        fl.append(f)
    fx.comm_init_peer(fl)
    for k in range(steps):
        fl[0].UpdateFrame(f32(fl[0].default_time_step()), k % 3)
        fl[0].Simulate(k % 3)
    fl[0].Synchronize()
    assert np.array_equal(gather(fl, fx.FIELD_VELOCITY, 1), ref.download(fx.FIELD_VELOCITY))
    assert np.array_equal(gather(fl, fx.FIELD_COLOR, 0), ref.download(fx.FIELD_COLOR))
    assert np.array_equal(gather(fl, fx.FIELD_PRESSURE, 0), ref.download(fx.FIELD_PRESSURE))


@pytest.mark.parametrize("dims", [(320, 320, 120), (224, 224, 210)])
@pytest.mark.parametrize("hj,iters,overlap", [(9, 23, 1), (7, 16, 2), (5, 11, 0), (9, 20, 2), (8, 40, 2), (8, 23, 2)])
def test_slabs_of_any_row_length_take_the_tiled_four_sweep_kernel(hj, iters, overlap, dims):
    """k_jacobi_strip4t (rows of 320 cells: two x tiles) on the trapezoid ranges of slab ranks: three slabs of 320 x 320 x 120, rounds of
    9 / 7 / 5 / 8 sweeps as 1 + 4 + 4, 1 + 1 + 1 + 4, 1 + 4, 4 + 4 (these rows have no three- or two-sweep kernel: the schedule composes a
    round from launches that exist); 224 x 224 x 210: rows below 256 cells -- one tile, its upper lanes switched off; bit-identical to one
    sweep per launch on the single domain"""
    ref = run_single(dims, 2, jacobi_iters=iters, jacobi_fuse=1)
    fl = run_slabs(dims, 2, 3, jacobi_iters=iters, halo_jacobi=hj, halo_advect=8, overlap=overlap)
    fl[0].timing_enable(True); fl[0].timing_read(True)
    fl[0].UpdateFrame(f32(fl[0].default_time_step()), 2)
    fl[0].Simulate(2)
    fl[0].Synchronize()
    t = fl[0].timing_read()
    assert t.jacobi_launches < t.jacobi_sweeps                       # fused launches took part
    ref.UpdateFrame(f32(ref.default_time_step()), 2)
    ref.Simulate(2)
    ref.Synchronize()
    assert np.array_equal(gather(fl, fx.FIELD_PRESSURE, 0), ref.download(fx.FIELD_PRESSURE))
    assert np.array_equal(gather(fl, fx.FIELD_VELOCITY, 1), ref.download(fx.FIELD_VELOCITY))
    assert np.abs(ref.download(fx.FIELD_PRESSURE)).max() > 0


@pytest.mark.parametrize("kernel", ["octet", "quad"])
@pytest.mark.parametrize("hj,iters,overlap", [(9, 23, 1), (7, 16, 1), (5, 11, 0), (9, 20, 2)])
def test_four_sweep_kernel_on_uneven_slab_ranges(hj, iters, overlap, kernel, knob):
    """k_jacobi_strip4o / k_jacobi_strip4q on the trapezoid ranges of slab ranks: three slabs of 256 x 256 x 470 (156 / 157 / 157 planes: above the 9.4 M cells
    from which fours are the default), rounds of 9 / 7 / 5 sweeps = 4 + 3 + 2, 4 + 3, 3 + 2 -- launch ranges that start and end inside the
    halo, chunks cut off by the first / last present plane (no fill), the global faces in the outer ranks; bit-identical to one sweep
    per launch on the single domain"""
    knob("STRIP4_OCTET", "1" if kernel == "octet" else "0")
    dims = (256, 256, 470)
    ref = run_single(dims, 2, jacobi_iters=iters, jacobi_fuse=1)
    fl = run_slabs(dims, 2, 3, jacobi_iters=iters, halo_jacobi=hj, halo_advect=8, overlap=overlap)
    assert np.array_equal(gather(fl, fx.FIELD_PRESSURE, 0), ref.download(fx.FIELD_PRESSURE))
    assert np.array_equal(gather(fl, fx.FIELD_VELOCITY, 1), ref.download(fx.FIELD_VELOCITY))
    assert np.abs(ref.download(fx.FIELD_PRESSURE)).max() > 0
