#!/usr/bin/env python3
"""Golden vectors from the reference's OWN shipped shader binaries.

Runs /root/reference/Bin/CS*.cso (the DXBC the reference executes) in the lock-step interpreter
tools/dxbc_interp.py on small seeded inputs and writes inputs + outputs to tests/golden/dxbc_*.npz.
Only runs in the authoring container (needs /root/reference); the fixtures are data (arrays in / arrays out).

    python tools/make_dxbc_golden.py            # regenerates every fixture (about a minute)

Resource bindings and constant-buffer layouts follow the reference's host code: Fluid.cpp:12-33 (CB structs),
:729-770 (SRV/UAV pairing), :825-908 (root constants of the three ray-march dispatches), LightProbeEZ.cpp:183-278.
Nothing in here calls the oracle or the HIP library.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import dxbc_interp as di  # noqa: E402

BIN = "/root/reference/Bin"
OUT = os.path.join(ROOT, "tests", "golden")
F32 = np.float32
U32 = np.uint32


def cb_floats(rows):
    a = np.zeros((len(rows), 4), F32)
    for i, r in enumerate(rows):
        a[i, :len(r)] = r
    return a.view(U32)


def divup(a, b):
    return (a + b - 1) // b


# ---------------------------------------------------------------------------------------------------------
# simulation
# ---------------------------------------------------------------------------------------------------------
def run_advect(vel, col, dt, address, fmt):
    """vel (3,Z,Y,X), col (Z,Y,X,4) -> (vel_out, col_out) through CSAdvect.cso (dispatch Fluid.cpp:374)"""
    _, Z, Y, X = vel.shape
    tv = di.Texture(np.moveaxis(vel, 0, -1), fmt)                 # Texture3D<float3> view of the RGBA16F texture
    tc = di.Texture(col, fmt)
    uv = di.Texture(np.zeros((Z, Y, X, 3), F32), fmt)
    uc = di.Texture(np.zeros((Z, Y, X, 4), F32), fmt)
    cb = {0: np.array([[np.float32(dt).view(U32), 12345, 0, 0]], U32)}    # CBSimulation {TimeStep, BaseSeed}
    di.run_shader(os.path.join(BIN, "CSAdvect.cso"), (divup(X, 8), divup(Y, 8), Z),
                  {"t0": tv, "t1": tc, "u0": uv, "u1": uc}, cb, {"s0": di.Sampler(address)})
    return np.moveaxis(uv.data, -1, 0).copy(), uc.data.copy()


def run_project(vel1, p, dt, fmt):
    """vel1 (3,Z,Y,X), pressure (Z,Y,X) -> (vel0, pressure) through CSProject3D/2D.cso (Fluid.cpp:394-408)"""
    _, Z, Y, X = vel1.shape
    tv = di.Texture(np.moveaxis(vel1, 0, -1), fmt)
    uv = di.Texture(np.zeros((Z, Y, X, 3), F32), fmt)
    up = di.Texture(p[..., None].copy(), "R32_FLOAT")
    cb = {0: np.array([[np.float32(dt).view(U32), 0, 0, 0]], U32)}
    if Z > 1:
        m = di.run_shader(os.path.join(BIN, "CSProject3D.cso"), (divup(X, 4), divup(Y, 4), divup(Z, 4)),
                          {"t0": tv, "u0": uv, "u1": up}, cb)
    else:
        m = di.run_shader(os.path.join(BIN, "CSProject2D.cso"), (divup(X, 8), divup(Y, 8), Z),
                          {"t0": tv, "u0": uv, "u1": up}, cb)
    return np.moveaxis(uv.data, -1, 0).copy(), up.data[..., 0].copy(), m.executed


def rand_state(X, Y, Z, seed, scale):
    rng = np.random.default_rng(seed)
    vel = (rng.standard_normal((3, Z, Y, X)) * scale).astype(F32)
    col = rng.random((Z, Y, X, 4)).astype(F32)
    p = (rng.standard_normal((Z, Y, X)) * 0.05).astype(F32)
    return vel, col, p


def make_sim():
    out = {}
    for tag, dims in (("3d", (16, 16, 8)), ("2d", (16, 16, 1))):
        X, Y, Z = dims
        dt = F32((2.0 if Z > 1 else 1.0) / Y)
        for fmt_tag, fmt in (("f32", "R32G32B32A32_FLOAT"), ("f16", "R16G16B16A16_FLOAT")):
            vel, col, p = rand_state(X, Y, Z, 101, 1.2)
            if fmt_tag == "f16":
                vel, col = di.half_round(vel), di.half_round(col)
            for address in ("CLAMP", "MIRROR"):
                vo, co = run_advect(vel, col, dt, address, fmt)
                k = "advect_%s_%s_%s" % (tag, fmt_tag, address.lower())
                out[k + "_vel_in"], out[k + "_col_in"], out[k + "_vel_out"], out[k + "_col_out"] = vel, col, vo, co
            v0, pp, n = run_project(vel, p, dt, fmt)
            k = "project_%s_%s" % (tag, fmt_tag)
            out[k + "_vel_in"], out[k + "_p_in"], out[k + "_vel_out"], out[k + "_p_out"] = vel, p, v0, pp
            print(k, "instructions executed:", n)
    # dt = 0: the projection block is skipped (CSProject3D.hlsl:88)
    vel, col, p = rand_state(16, 16, 8, 102, 0.5)
    v0, pp, _ = run_project(vel, p, 0.0, "R32G32B32A32_FLOAT")
    out["project_3d_paused_vel_in"], out["project_3d_paused_p_in"] = vel, p
    out["project_3d_paused_vel_out"], out["project_3d_paused_p_out"] = v0, pp

    # ---- 4-step rollout from the zero state with the reference's own formats (RGBA16F fields, R32F pressure,
    #      ITER 64 + early-out) and host sequencing (Fluid.cpp:345,360-384): advect vel0->vel1, colour[!p]->colour[p];
    #      project vel1->vel0, pressure in place
    for tag, dims, address in (("3d", (16, 16, 16), "MIRROR"), ("3d_ez", (16, 16, 16), "CLAMP"), ("2d", (32, 32, 1), "CLAMP")):
        X, Y, Z = dims
        dt = F32((2.0 if Z > 1 else 1.0) / Y)
        vel0 = np.zeros((3, Z, Y, X), F32)
        cols = [np.zeros((Z, Y, X, 4), F32), np.zeros((Z, Y, X, 4), F32)]
        p = np.zeros((Z, Y, X), F32)
        parity = 0
        for step in range(4):
            parity ^= 1
            vel1, cols[parity] = run_advect(vel0, cols[1 - parity], dt, address, "R16G16B16A16_FLOAT")
            vel0, p, _ = run_project(vel1, p, dt, "R16G16B16A16_FLOAT")
            out["rollout_%s_step%d_vel" % (tag, step + 1)] = vel0
            out["rollout_%s_step%d_col" % (tag, step + 1)] = cols[parity]
            out["rollout_%s_step%d_p" % (tag, step + 1)] = p
        print("rollout", tag, "max|u|", float(np.abs(vel0).max()), "sum alpha", float(cols[parity][..., 3].sum()))
    np.savez_compressed(os.path.join(OUT, "dxbc_sim.npz"), **out)


# ---------------------------------------------------------------------------------------------------------
# ray march
# ---------------------------------------------------------------------------------------------------------
def frame_constants(X, vw, vh):
    """what Fluid::UpdateFrame writes (Fluid.cpp:296-334) for the demo's default camera (FluidX12.cpp:243-253),
    in float64 numpy then rounded -- these are INPUTS of the golden vectors and are stored with them."""
    eye = np.array([4.0, 16.0, -40.0])
    z = -eye / np.linalg.norm(eye)
    x = np.cross([0, 1.0, 0], z); x /= np.linalg.norm(x)
    y = np.cross(z, x)
    view = np.eye(4); view[:3, 0], view[:3, 1], view[:3, 2] = x, y, z
    view[3, :3] = [-x @ eye, -y @ eye, -z @ eye]
    h = 1.0 / np.tan(np.pi / 8); w = h / (vw / vh); q = 1000.0 / 999.0
    proj = np.zeros((4, 4)); proj[0, 0], proj[1, 1], proj[2, 2], proj[2, 3], proj[3, 2] = w, h, q, 1.0, -q
    world = np.diag([10.0, 10.0, 10.0, 1.0])
    wvp = world @ view @ proj
    rows = np.zeros((14, 4))
    rows[0:4] = np.linalg.inv(wvp).T
    rows[4:8] = wvp.T
    rows[8:11] = np.linalg.inv(world).T[:3]
    rows[11:14] = world.T[:3]
    cb0 = rows.astype(F32)
    pi = np.float32(3.141592654)
    cb1 = np.array([[4, 16, -40, 1], [75, 75, -75, 1], [1, .7, .3, pi * F32(3)], [1, 1, 1, pi * F32(1.5)]], F32)
    return cb0, cb1


def smoke_volume(X, seed):
    rng = np.random.default_rng(seed)
    z, y, x = np.meshgrid(*(np.arange(X),) * 3, indexing="ij")
    blob = np.exp(-(((x - X * .55) ** 2 + (y - X * .45) ** 2 + (z - X * .5) ** 2) / (X * .28) ** 2))
    col = (blob[..., None] * (0.35 + 0.65 * rng.random((X, X, X, 4))) * np.array([.4, .6, .9, .8])).astype(F32)
    col[col[..., 3] < 0.02] = 0
    return di.half_round(col)


def make_render():
    out = {}
    X, S = 16, 16
    cb0, cb1 = frame_constants(X, 640, 480)
    col = smoke_volume(X, 7)
    sh = (np.random.default_rng(8).random((9, 3)) * np.array([[2.0]] + [[0.6]] * 8)).astype(F32)
    out["color"], out["cb_per_object"], out["cb_per_frame"], out["sh"] = col, cb0, cb1, sh
    mask = 0x1B
    for has_sh in (0, 1):
        # light pass (Fluid.cpp:857-878): cb2 = {maxLightSamples, hasSH}
        nl = 16
        lm = di.Texture(np.zeros((X, X, X, 3), F32), "R11G11B10_FLOAT")
        di.run_shader(os.path.join(BIN, "CSRayMarchL.cso"), (X // 4, X // 4, X // 4),
                      {"t0": di.Texture(col), "t1": di.Structured(9, 12, sh), "u0": lm},
                      {0: cb0.view(U32), 1: cb1.view(U32), 2: np.array([[nl, has_sh, 0, 0]], U32)}, {"s0": di.Sampler("CLAMP")})
        out["lightmap_sh%d" % has_sh] = lm.data.copy()
        # view pass with the light map (Fluid.cpp:880-908): cb2 = {raySampleCount}, cb3 = mask
        ns = 24
        cube = di.Texture(np.zeros((6, S, S, 4), F32), "R8G8B8A8_UNORM")
        di.run_shader(os.path.join(BIN, "CSRayMarchV.cso"), (S // 8, S // 8, 6),
                      {"t0": di.Texture(col), "t1": di.Texture(lm.data), "u0": cube},
                      {0: cb0.view(U32), 1: cb1.view(U32), 2: np.array([[ns, 0, 0, 0]], U32), 3: np.array([[mask, 0, 0, 0]], U32)},
                      {"s0": di.Sampler("CLAMP")})
        out["cube_separate_sh%d" % has_sh] = np.rint(cube.data * 255).astype(np.uint8)
        # merged march (Fluid.cpp:825-855): cb2 = {raySampleCount, hasSH, maxLightSamples}
        cube = di.Texture(np.zeros((6, S, S, 4), F32), "R8G8B8A8_UNORM")
        di.run_shader(os.path.join(BIN, "CSRayMarch.cso"), (S // 8, S // 8, 6),
                      {"t0": di.Texture(col), "t1": di.Structured(9, 12, sh), "u0": cube},
                      {0: cb0.view(U32), 1: cb1.view(U32), 2: np.array([[ns, has_sh, 8, 0]], U32), 3: np.array([[mask, 0, 0, 0]], U32)},
                      {"s0": di.Sampler("CLAMP")})
        out["cube_merged_sh%d" % has_sh] = np.rint(cube.data * 255).astype(np.uint8)
        print("render sh=%d: lightmap range %.3f..%.3f, cube alpha max %d" % (
            has_sh, lm.data.min(), lm.data.max(), out["cube_separate_sh%d" % has_sh][..., 3].max()))
    out["params"] = np.array([X, S, 16, 24, 8, mask, 640, 480], np.int64)     # grid, cube, light samples, view, merged light, mask, vp
    np.savez_compressed(os.path.join(OUT, "dxbc_render.npz"), **out)


# ---------------------------------------------------------------------------------------------------------
# cube map -> screen resolve (row f-1): PSRayCastCube.cso per screen pixel
# ---------------------------------------------------------------------------------------------------------
def make_resolve():
    """inputs: a cube map (the view pass's own golden output, and a fully random one that exercises every seamless edge),
    the frame constants and the screen-quad interpolant UV = (pixel + 0.5) / size (VSScreenQuad.hlsl:17-26);
    outputs: SV_TARGET (premultiplied RGBA, fp32) and the discard mask."""
    out = {}
    X = 16
    W, H = 160, 120                                                   # same aspect as the 640x480 of the frame constants
    cb0, cb1 = frame_constants(X, 640, 480)
    out["cb_per_object"], out["cb_per_frame"] = cb0, cb1
    py, px = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    uv = np.zeros((H * W, 4), F32)
    uv[:, 0] = ((px.ravel().astype(F32) + F32(0.5)) / F32(W)).astype(F32)
    uv[:, 1] = ((py.ravel().astype(F32) + F32(0.5)) / F32(H)).astype(F32)
    rendered = np.load(os.path.join(OUT, "dxbc_render.npz"))["cube_separate_sh0"]
    rng = np.random.default_rng(11)
    random8 = rng.integers(0, 256, (6, 8, 8, 4), dtype=np.uint8)
    random8[..., 3] = np.maximum(random8[..., 3], 1)
    for name, cube in (("rendered16", rendered), ("random8", random8)):
        m = di.run_pixel_shader(os.path.join(BIN, "PSRayCastCube.cso"), {1: uv}, {"t0": di.CubeSeamless(cube)},
                                {0: cb0.view(U32), 1: cb1.view(U32)}, {"s0": di.Sampler("CLAMP")})
        res = m.outputs[0].view(F32).reshape(H, W, 4).copy()
        disc = m.discarded.reshape(H, W)
        res[disc] = 0
        out["cube_" + name], out["target_" + name], out["discard_" + name] = cube, res, disc
        print("resolve %s: %d of %d pixels covered, alpha max %.3f" % (name, (~disc).sum(), disc.size, res[..., 3].max()))
    out["params"] = np.array([W, H, 640, 480], np.int64)
    np.savez_compressed(os.path.join(OUT, "dxbc_resolve.npz"), **out)


# ---------------------------------------------------------------------------------------------------------
# direct screen-space march (row f-2): PSRayCast.cso / PSRayCastV.cso per screen pixel
# ---------------------------------------------------------------------------------------------------------
def make_direct():
    """the volume, SH and light maps of the cube-map goldens marched per screen pixel instead of per cube texel:
    rayCastVDirect binds {colour, light map}, cb2 = {raySampleCount} (Fluid.cpp:953-972); rayCastDirect binds
    {colour, SH buffer}, cb2 = {maxRaySamples, hasSH, maxLightSamples} (Fluid.cpp:932-951)."""
    ren = np.load(os.path.join(OUT, "dxbc_render.npz"))
    col, sh, cb0, cb1 = ren["color"], ren["sh"], ren["cb_per_object"], ren["cb_per_frame"]
    W, H = 96, 72
    py, px = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    uv = np.zeros((H * W, 4), F32)
    uv[:, 0] = ((px.ravel().astype(F32) + F32(0.5)) / F32(W)).astype(F32)
    uv[:, 1] = ((py.ravel().astype(F32) + F32(0.5)) / F32(H)).astype(F32)
    out = {"params": np.array([W, H, 24, 8, 640, 480], np.int64)}       # target size, view samples, merged light samples, vp
    ns, nml = 24, 8
    for has_sh in (0, 1):
        lm = ren["lightmap_sh%d" % has_sh]
        m = di.run_pixel_shader(os.path.join(BIN, "PSRayCastV.cso"), {1: uv},
                                {"t0": di.Texture(col), "t1": di.Texture(lm)},
                                {0: cb0.view(U32), 1: cb1.view(U32), 2: np.array([[ns, 0, 0, 0]], U32)}, {"s0": di.Sampler("CLAMP")})
        res = m.outputs[0].view(F32).reshape(H, W, 4).copy(); res[m.discarded.reshape(H, W)] = 0
        out["separate_sh%d" % has_sh], out["discard_separate_sh%d" % has_sh] = res, m.discarded.reshape(H, W)
        m = di.run_pixel_shader(os.path.join(BIN, "PSRayCast.cso"), {1: uv},
                                {"t0": di.Texture(col), "t1": di.Structured(9, 12, sh)},
                                {0: cb0.view(U32), 1: cb1.view(U32), 2: np.array([[ns, has_sh, nml, 0]], U32)}, {"s0": di.Sampler("CLAMP")})
        res = m.outputs[0].view(F32).reshape(H, W, 4).copy(); res[m.discarded.reshape(H, W)] = 0
        out["merged_sh%d" % has_sh], out["discard_merged_sh%d" % has_sh] = res, m.discarded.reshape(H, W)
        print("direct sh=%d: separate alpha max %.3f (%d px covered), merged alpha max %.3f" % (
            has_sh, out["separate_sh%d" % has_sh][..., 3].max(), (~m.discarded).sum(), res[..., 3].max()))
    # 2-D visualiser (PSVisualizeColor.cso, Fluid.cpp:811-823): a 24 x 24 fp16-exact colour field on a 40 x 30 target
    rng = np.random.default_rng(12)
    col2d = di.half_round((rng.random((1, 24, 24, 4)) * np.array([1.5, 1.0, 0.5, 1.0])).astype(F32))
    W2, H2 = 40, 30
    py, px = np.meshgrid(np.arange(H2), np.arange(W2), indexing="ij")
    uv2 = np.zeros((H2 * W2, 4), F32)
    uv2[:, 0] = ((px.ravel().astype(F32) + F32(0.5)) / F32(W2)).astype(F32)
    uv2[:, 1] = ((py.ravel().astype(F32) + F32(0.5)) / F32(H2)).astype(F32)
    m = di.run_pixel_shader(os.path.join(BIN, "PSVisualizeColor.cso"), {1: uv2}, {"t0": di.Texture(col2d)}, {}, {"s0": di.Sampler("CLAMP")})
    out["visualize_color"], out["visualize_target"] = col2d, m.outputs[0].view(F32).reshape(H2, W2, 4).copy()
    np.savez_compressed(os.path.join(OUT, "dxbc_direct.npz"), **out)


# ---------------------------------------------------------------------------------------------------------
# sky pass of the light probe: PSEnvironment.cso per screen pixel (LightProbe::RenderEnvironment, LightProbe.cpp:85-97)
# ---------------------------------------------------------------------------------------------------------
def make_environment():
    """cbPerFrame = {eyePt, screenToWorld = transpose(inverse(view * proj))} (LightProbe.cpp:70-76) for the demo's default camera,
    a random HDR float cube as g_txEnv; output = SV_TARGET (rgb, alpha 0)."""
    eye = np.array([4.0, 16.0, -40.0])
    z = -eye / np.linalg.norm(eye)
    x = np.cross([0, 1.0, 0], z); x /= np.linalg.norm(x)
    y = np.cross(z, x)
    view = np.eye(4); view[:3, 0], view[:3, 1], view[:3, 2] = x, y, z
    view[3, :3] = [-x @ eye, -y @ eye, -z @ eye]
    vw, vh = 640, 480
    h = 1.0 / np.tan(np.pi / 8); w = h / (vw / vh); q = 1000.0 / 999.0
    proj = np.zeros((4, 4)); proj[0, 0], proj[1, 1], proj[2, 2], proj[2, 3], proj[3, 2] = w, h, q, 1.0, -q
    cb = np.zeros((5, 4), F32)
    cb[0, :3] = eye
    cb[1:5] = np.linalg.inv(view @ proj).T.astype(F32)
    rng = np.random.default_rng(21)
    cube = (rng.random((6, 8, 8, 3)) ** 4 * 30.0).astype(F32)          # HDR-ish: mostly dim, a few bright texels
    W, H = 128, 96
    py, px = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    uv = np.zeros((H * W, 4), F32)
    uv[:, 0] = ((px.ravel().astype(F32) + F32(0.5)) / F32(W)).astype(F32)
    uv[:, 1] = ((py.ravel().astype(F32) + F32(0.5)) / F32(H)).astype(F32)
    m = di.run_pixel_shader(os.path.join(BIN, "PSEnvironment.cso"), {1: uv}, {"t0": di.CubeSeamless(cube)}, {0: cb.view(U32)},
                            {"s0": di.Sampler("WRAP")})
    out = {"cube": cube, "cb": cb, "target": m.outputs[0].view(F32).reshape(H, W, 4).copy(), "params": np.array([W, H, vw, vh], np.int64)}
    print("environment: target range %.3f .. %.3f, alpha max %.1f" % (out["target"][..., :3].min(), out["target"][..., :3].max(), out["target"][..., 3].max()))
    np.savez_compressed(os.path.join(OUT, "dxbc_env.npz"), **out)


# ---------------------------------------------------------------------------------------------------------
# spherical harmonics
# ---------------------------------------------------------------------------------------------------------
def make_sh():
    out = {}
    N, order = 16, 3
    rng = np.random.default_rng(9)
    cube = (rng.random((6, N, N, 3)) * np.array([1.0, 0.8, 0.6])).astype(F32)
    total = 6 * N * N
    groups = divup(total, 32)
    sh0, w0 = di.Structured(groups * 9, 12), di.Structured(groups, 4)
    sums = divup(groups, 32)
    sh1, w1 = di.Structured(max(sums, 1) * 9 + 9, 12), di.Structured(max(sums, 1) + 1, 4)
    di.run_shader(os.path.join(BIN, "CSSHCubeMap.cso"), (groups, 1, 1),
                  {"t0": di.CubePoint(cube), "u0": sh0, "u1": w0}, {0: np.array([[order, N, 0, 0]], U32)}, {"s0": di.Sampler("WRAP")})
    out["partials_sh"], out["partials_w"] = sh0.words.view(F32).copy(), w0.words.view(F32).copy()
    S, W = [sh0, sh1], [w0, w1]
    src, n = 0, groups
    while n > 1:                                                       # LightProbeEZ.cpp:213-252 with the intended per-pass count
        di.run_shader(os.path.join(BIN, "CSSHSum.cso"), (divup(n, 32), order * order, 1),
                      {"t0": S[src], "t1": W[src], "u0": S[src ^ 1], "u1": W[src ^ 1]}, {0: np.array([[order, n, 0, 0]], U32)})
        src ^= 1
        n = divup(n, 32)
    res = di.Structured(32, 12)
    di.run_shader(os.path.join(BIN, "CSSHNormalize.cso"), (1, 1, 1), {"t0": S[src], "t1": W[src], "u0": res}, {})
    out["cube"], out["sh"] = cube, res.words.view(F32)[:9].copy()
    print("SH[0] =", out["sh"][0], " weight sum ->", float(W[src].words.view(F32)[0, 0]), "(4 pi = 12.566)")
    np.savez_compressed(os.path.join(OUT, "dxbc_sh.npz"), **out)


# ---------------------------------------------------------------------------------------------------------
# round 5: where the pin was blind -- over-covered grids, a longer rollout, a coarser cube mip, the real SH size and the as-shipped
# SH reduction (tests/golden/dxbc_wide.npz)
# ---------------------------------------------------------------------------------------------------------
def run_sh_chain(cube, order=3, as_shipped=False):
    """CSSHCubeMap -> CSSHSum x k -> CSSHNormalize exactly as LightProbeEZ.cpp:183-278 issues them.  as_shipped: every CSSHSum pass
    reads constant-buffer slice 0 (LightProbeEZ.cpp:245-246 binds the buffer's first slice for all passes), i.e. g_pixelCount of the
    FIRST pass; otherwise each pass sees its own element count (what the host code fills into the slices, :101-102)."""
    N = cube.shape[1]
    total = 6 * N * N
    groups = divup(total, 32)
    sums = divup(groups, 32)
    sh0, w0 = di.Structured(groups * 9, 12), di.Structured(groups, 4)
    sh1, w1 = di.Structured(max(sums, 1) * 9 + 9, 12), di.Structured(max(sums, 1) + 1, 4)
    di.run_shader(os.path.join(BIN, "CSSHCubeMap.cso"), (groups, 1, 1),
                  {"t0": di.CubePoint(cube), "u0": sh0, "u1": w0}, {0: np.array([[order, N, 0, 0]], U32)}, {"s0": di.Sampler("WRAP")})
    S, W = [sh0, sh1], [w0, w1]
    src, n, passes = 0, groups, 0
    while n > 1:
        count = groups if as_shipped else n
        di.run_shader(os.path.join(BIN, "CSSHSum.cso"), (divup(n, 32), order * order, 1),
                      {"t0": S[src], "t1": W[src], "u0": S[src ^ 1], "u1": W[src ^ 1]}, {0: np.array([[order, count, 0, 0]], U32)})
        src ^= 1
        n = divup(n, 32)
        passes += 1
    res = di.Structured(32, 12)
    di.run_shader(os.path.join(BIN, "CSSHNormalize.cso"), (1, 1, 1), {"t0": S[src], "t1": W[src], "u0": res}, {})
    return res.words.view(F32)[:9].copy(), passes


def make_wide():
    out = {}
    # ---- grids the dispatch OVER-COVERS (the reference launches ceil(N / 8) or ceil(N / 4) groups and relies on out-of-range loads
    #      returning 0 and out-of-range stores being dropped: CSProject3D.hlsl:68-113 at the 150^3 of Bin/FluidGI.bat)
    for tag, dims in (("3d", (20, 20, 10)), ("2d", (12, 12, 1))):
        X, Y, Z = dims
        dt = F32((2.0 if Z > 1 else 1.0) / Y)
        vel, col, p = rand_state(X, Y, Z, 201, 1.2)
        for address in ("CLAMP", "MIRROR"):
            vo, co = run_advect(vel, col, dt, address, "R32G32B32A32_FLOAT")
            k = "advect_%s_%s" % (tag, address.lower())
            out[k + "_vel_in"], out[k + "_col_in"], out[k + "_vel_out"], out[k + "_col_out"] = vel, col, vo, co
        v0, pp, n = run_project(vel, p, dt, "R32G32B32A32_FLOAT")
        k = "project_%s" % tag
        out[k + "_vel_in"], out[k + "_p_in"], out[k + "_vel_out"], out[k + "_p_out"] = vel, p, v0, pp
        print("wide", k, dims, "instructions executed:", n)
    # ---- 8 frames of the reference's own configuration (class Fluid: MIRROR, RGBA16F, the 64-sweep early-out loop) on such a grid
    X, Y, Z = 20, 20, 12
    dt = F32(2.0 / Y)
    vel0 = np.zeros((3, Z, Y, X), F32)
    cols = [np.zeros((Z, Y, X, 4), F32), np.zeros((Z, Y, X, 4), F32)]
    p = np.zeros((Z, Y, X), F32)
    parity = 0
    for step in range(8):
        parity ^= 1
        vel1, cols[parity] = run_advect(vel0, cols[1 - parity], dt, "MIRROR", "R16G16B16A16_FLOAT")
        vel0, p, _ = run_project(vel1, p, dt, "R16G16B16A16_FLOAT")
        if step + 1 in (2, 5, 8):
            out["rollout8_step%d_vel" % (step + 1)] = vel0
            out["rollout8_step%d_col" % (step + 1)] = cols[parity]
            out["rollout8_step%d_p" % (step + 1)] = p
    print("wide rollout 20x20x12, 8 frames: max|u|", float(np.abs(vel0).max()), "sum alpha", float(cols[parity][..., 3].sum()))
    # ---- light volume on a grid of 18^3 voxels (ceil(18 / 4) = 5 groups per axis: two idle threads per row) and the view march into a
    #      cube map one mip BELOW the grid (8^2 texels for 16^3: cube LOD 1, Fluid.cpp:324-333)
    X = 18
    cb0, cb1 = frame_constants(X, 640, 480)
    col = smoke_volume(X, 17)
    lm = di.Texture(np.zeros((X, X, X, 3), F32), "R11G11B10_FLOAT")
    di.run_shader(os.path.join(BIN, "CSRayMarchL.cso"), (divup(X, 4),) * 3,
                  {"t0": di.Texture(col), "t1": di.Structured(9, 12, np.zeros((9, 3), F32)), "u0": lm},
                  {0: cb0.view(U32), 1: cb1.view(U32), 2: np.array([[16, 0, 0, 0]], U32)}, {"s0": di.Sampler("CLAMP")})
    out["light18_color"], out["light18_cb_per_object"], out["light18_cb_per_frame"], out["light18_lightmap"] = col, cb0, cb1, lm.data.copy()
    print("wide light 18^3: lightmap range %.3f..%.3f" % (lm.data.min(), lm.data.max()))
    ren = np.load(os.path.join(OUT, "dxbc_render.npz"))
    X, S = 16, 8
    NS1 = 7                                                            # what Fluid::UpdateFrame derives for the 20 x 15 viewport that gives this volume cube LOD 1
    cb0, cb1 = ren["cb_per_object"], ren["cb_per_frame"]
    out["cube_lod1_params"] = np.array([X, S, 16, NS1, 8, 0x1B, 20, 15], np.int64)   # grid, cube, light samples, view, merged light, mask, viewport
    cube = di.Texture(np.zeros((6, S, S, 4), F32), "R8G8B8A8_UNORM")
    di.run_shader(os.path.join(BIN, "CSRayMarchV.cso"), (S // 8, S // 8, 6),
                  {"t0": di.Texture(ren["color"]), "t1": di.Texture(ren["lightmap_sh0"]), "u0": cube},
                  {0: cb0.view(U32), 1: cb1.view(U32), 2: np.array([[NS1, 0, 0, 0]], U32), 3: np.array([[0x1B, 0, 0, 0]], U32)},
                  {"s0": di.Sampler("CLAMP")})
    out["cube_lod1_separate"] = np.rint(cube.data * 255).astype(np.uint8)
    cube = di.Texture(np.zeros((6, S, S, 4), F32), "R8G8B8A8_UNORM")
    di.run_shader(os.path.join(BIN, "CSRayMarch.cso"), (S // 8, S // 8, 6),
                  {"t0": di.Texture(ren["color"]), "t1": di.Structured(9, 12, ren["sh"]), "u0": cube},
                  {0: cb0.view(U32), 1: cb1.view(U32), 2: np.array([[NS1, 0, 8, 0]], U32), 3: np.array([[0x1B, 0, 0, 0]], U32)},
                  {"s0": di.Sampler("CLAMP")})
    out["cube_lod1_merged"] = np.rint(cube.data * 255).astype(np.uint8)
    print("wide cube LOD 1: alpha max", out["cube_lod1_separate"][..., 3].max(), out["cube_lod1_merged"][..., 3].max())
    # ---- the SH chain at the reference's real size (SH_TEX_SIZE 256: 12288 partials, three CSSHSum passes), as intended and AS SHIPPED
    rng = np.random.default_rng(19)
    N = 256
    base = (rng.random((6, 32, 32, 3)) * np.array([1.0, 0.8, 0.6])).astype(F32)   # HDR-ish radiance on a 32^2 pattern ...
    base[2, 8:16, 8:16] *= F32(20.0)                                  # ... with a bright window in one face
    cube = np.repeat(np.repeat(base, N // 32, axis=1), N // 32, axis=2)   # every pattern texel covers 8 x 8 cube texels: the fixture keeps the pattern
    sh_i, passes = run_sh_chain(cube, 3, False)
    sh_s, _ = run_sh_chain(cube, 3, True)
    out["sh256_base32"] = base
    out["sh256_intended"], out["sh256_as_shipped"], out["sh256_passes"] = sh_i, sh_s, np.array([passes], np.int64)
    dev = np.abs(sh_s - sh_i).max() / np.abs(sh_i).max()
    print("wide SH %d^2: %d CSSHSum passes; as-shipped vs intended: max |d| / max |sh| = %.3e" % (N, passes, dev))
    print("   intended  ", sh_i.ravel()[:6])
    print("   as shipped", sh_s.ravel()[:6])
    np.savez_compressed(os.path.join(OUT, "dxbc_wide.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["sim", "render", "sh", "resolve", "direct", "env", "wide"]
    if "sim" in which:
        make_sim()
    if "render" in which:
        make_render()
    if "sh" in which:
        make_sh()
    if "resolve" in which:
        make_resolve()
    if "direct" in which:
        make_direct()
    if "env" in which:
        make_environment()
    if "wide" in which:
        make_wide()
