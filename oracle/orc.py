"""ctypes binding of the CPU oracle (oracle/liborc.so).

TEST INFRASTRUCTURE ONLY -- may be imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package (fluidx12_amd/).  Builds liborc.so with `make`
when it is missing or older than its sources (gcc is present on the GPU box too).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liborc.so")
_SRCS = ["orc_sim.cpp", "orc_render.cpp", "orc_host.cpp", "orc_sh.cpp", "orc_resolve.cpp", "orc_bc6h.cpp", "orc_common.h", "fx_oracle.h",
         "Makefile"]


def build(force=False):
    stale = force or not os.path.exists(_LIB)
    if not stale:
        t = os.path.getmtime(_LIB)
        stale = any(os.path.getmtime(os.path.join(_HERE, s)) > t for s in _SRCS)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liborc.so"], stdout=subprocess.DEVNULL)
    return _LIB


class Frame(C.Structure):
    _fields_ = [("world_i", C.c_float * 12), ("world", C.c_float * 12), ("eye_pt", C.c_float * 3),
                ("light_pt", C.c_float * 3), ("light_color", C.c_float * 4), ("ambient", C.c_float * 4),
                ("sh", C.c_float * 27)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        fp = C.POINTER(C.c_float)
        u8p = C.POINTER(C.c_uint8)
        i, f, u = C.c_int, C.c_float, C.c_uint32
        _lib.orc_advect.argtypes = [fp, fp, fp, fp, i, i, i, f, i, i]
        _lib.orc_divergence.argtypes = [fp, fp, i, i, i]
        _lib.orc_jacobi_sweep.argtypes = [fp, fp, fp, u8p, i, i, i]
        _lib.orc_jacobi_sweep.restype = C.c_longlong
        _lib.orc_jacobi.argtypes = [fp, fp, fp, i, i, i, i, i]
        _lib.orc_jacobi.restype = i
        _lib.orc_project.argtypes = [fp, fp, fp, i, i, i, i]
        _lib.orc_simulate.argtypes = [fp, fp, fp, fp, fp, fp, fp, i, i, i, f, i, i, i, i]
        _lib.orc_quantize_half.argtypes = [fp, fp, C.c_longlong]
        _lib.orc_f32_to_f16.argtypes = [f]
        _lib.orc_f32_to_f16.restype = C.c_uint16
        _lib.orc_f16_to_f32.argtypes = [C.c_uint16]
        _lib.orc_f16_to_f32.restype = f
        _lib.orc_look_at_lh.argtypes = [fp, fp, fp, fp]
        _lib.orc_perspective_fov_lh.argtypes = [f, f, f, f, fp]
        _lib.orc_update_frame.argtypes = [fp, fp, fp, u, u, u, u, C.POINTER(Frame), C.POINTER(u), C.POINTER(u),
                                          C.POINTER(u), fp]
        _lib.orc_raymarch_light.argtypes = [fp, fp, i, i, i, C.POINTER(Frame), u, i, i]
        _lib.orc_raymarch_view.argtypes = [fp, fp, i, i, i, C.POINTER(Frame), i, u, u, u, i, i, fp, u8p]
        _lib.orc_raycast_direct.argtypes = [fp, fp, i, i, i, C.POINTER(Frame), fp, i, i, u, u, i, i, fp, u8p]
        _lib.orc_visualize_color.argtypes = [fp, i, i, i, i, fp]
        _lib.orc_environment.argtypes = [fp, i, fp, fp, i, i, fp]
        _lib.orc_bc6h_decode_block.argtypes = [u8p, C.POINTER(C.c_uint16)]
        _lib.orc_dds_bc6h_cube_face.argtypes = [u8p, C.c_size_t, i, i, fp, C.POINTER(C.c_int)]
        _lib.orc_world_view_proj_inverse.argtypes = [fp, fp, fp]
        _lib.orc_resolve_cube.argtypes = [u8p, i, C.POINTER(Frame), fp, i, i, fp, u8p]
        _lib.orc_blend_premultiplied.argtypes = [fp, u8p, u8p, i, i]
        _lib.orc_pack_r11g11b10.argtypes = [f, f, f]
        _lib.orc_pack_r11g11b10.restype = u
        _lib.orc_unpack_r11g11b10.argtypes = [u, fp]
        _lib.orc_sh_transform.argtypes = [fp, i, fp, i]
        _lib.orc_sh_irradiance.argtypes = [fp, fp, fp]
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


# ------------------------------------------------------------------------------------------------
# simulation
# ------------------------------------------------------------------------------------------------
def advect(vel, col, dt, address=0, half=False):
    """vel (3,Z,Y,X), col (Z,Y,X,4) -> (vel_out, col_out)"""
    vel, col = _f32(vel), _f32(col)
    _, Z, Y, X = vel.shape
    vo, co = np.empty_like(vel), np.empty_like(col)
    lib().orc_advect(_fp(vel), _fp(col), _fp(vo), _fp(co), X, Y, Z, dt, address, int(half))
    return vo, co


def divergence(vel):
    vel = _f32(vel)
    _, Z, Y, X = vel.shape
    b = np.empty((Z, Y, X), np.float32)
    lib().orc_divergence(_fp(vel), _fp(b), X, Y, Z)
    return b


def jacobi(p, b, iters, mode=0):
    """returns (p_out, sweeps_executed)"""
    p = _f32(p).copy()
    b = _f32(b)
    Z, Y, X = p.shape
    tmp = np.empty_like(p)
    k = lib().orc_jacobi(_fp(p), _fp(b), _fp(tmp), X, Y, Z, iters, mode)
    return p, k


def project(vel, p, half=False):
    vel, p = _f32(vel), _f32(p)
    _, Z, Y, X = vel.shape
    out = np.empty_like(vel)
    lib().orc_project(_fp(vel), _fp(p), _fp(out), X, Y, Z, int(half))
    return out


class Sim:
    """Host-side state of the oracle: mirrors Fluid's resources and ping-pong rule
    (Fluid.cpp:204-221, 345, 360-384)."""

    def __init__(self, X, Y, Z, iters=40, mode=0, address=0, half=False):
        self.X, self.Y, self.Z = X, Y, Z
        self.iters, self.mode, self.address, self.half = iters, mode, address, half
        self.vel = [np.zeros((3, Z, Y, X), np.float32) for _ in range(2)]
        self.col = [np.zeros((Z, Y, X, 4), np.float32) for _ in range(2)]
        self.p = np.zeros((Z, Y, X), np.float32)
        self.b = np.zeros((Z, Y, X), np.float32)
        self.tmp = np.zeros((Z, Y, X), np.float32)
        self.parity = 0

    def default_dt(self):
        return (2.0 if self.Z > 1 else 1.0) / self.Y        # FluidX12.cpp:266

    def step(self, dt=None):
        dt = np.float32(self.default_dt() if dt is None else dt)
        if dt > 0:
            self.parity ^= 1                                 # UpdateFrame, Fluid.cpp:345
        p = self.parity
        lib().orc_simulate(_fp(self.vel[0]), _fp(self.vel[1]), _fp(self.col[1 - p]), _fp(self.col[p]),
                           _fp(self.p), _fp(self.b), _fp(self.tmp), self.X, self.Y, self.Z, float(dt),
                           self.iters, self.mode, self.address, int(self.half))

    @property
    def velocity(self):
        return self.vel[0]

    @property
    def color(self):
        return self.col[self.parity]


# ------------------------------------------------------------------------------------------------
# host rules / rendering
# ------------------------------------------------------------------------------------------------
def default_camera(width, height):
    """FluidX12.cpp:243-253: eye (4,16,-40) -> origin, up +y, FOV pi/4, z 1..1000."""
    view = np.empty(16, np.float32)
    proj = np.empty(16, np.float32)
    eye = np.array([4.0, 16.0, -40.0], np.float32)
    lib().orc_look_at_lh(_fp(eye), _fp(np.zeros(3, np.float32)), _fp(np.array([0, 1, 0], np.float32)), _fp(view))
    lib().orc_perspective_fov_lh(np.float32(np.pi) / np.float32(4), width / float(height), 1.0, 1000.0, _fp(proj))
    return view, proj, eye


def update_frame(view, proj, eye, vw, vh, grid_x, max_ray_samples=192):
    fr = Frame()
    lod, rs, mask = C.c_uint32(), C.c_uint32(), C.c_uint32()
    edge = C.c_float()
    view, proj, eye = _f32(view), _f32(proj), _f32(eye)
    lib().orc_update_frame(_fp(view), _fp(proj), _fp(eye), vw, vh, grid_x, max_ray_samples, C.byref(fr),
                           C.byref(lod), C.byref(rs), C.byref(mask), C.byref(edge))
    return fr, lod.value, rs.value, mask.value, edge.value


def raymarch_light(col, frame, num_samples=64, has_sh=False, light_fmt=2):
    col = _f32(col)
    Z, Y, X, _ = col.shape
    lm = np.empty((Z, Y, X, 3), np.float32)
    lib().orc_raymarch_light(_fp(col), _fp(lm), X, Y, Z, C.byref(frame), num_samples, int(has_sh), light_fmt)
    return lm


def raymarch_view(col, lightmap, frame, size, mask, num_samples, num_light_samples=64, has_sh=False, separate=True):
    col = _f32(col)
    Z, Y, X, _ = col.shape
    cf = np.zeros((6, size, size, 4), np.float32)
    cu = np.zeros((6, size, size, 4), np.uint8)
    lmp = _fp(_f32(lightmap)) if lightmap is not None else None
    lib().orc_raymarch_view(_fp(col), lmp, X, Y, Z, C.byref(frame), size, mask, num_samples, num_light_samples,
                            int(has_sh), int(separate), _fp(cf), cu.ctypes.data_as(C.POINTER(C.c_uint8)))
    return cf, cu


def raycast_direct(col, lightmap, frame, wvp_i, width, height, num_samples, num_light_samples=64, has_sh=False, separate=True):
    """PSRayCast / PSRayCastV per screen pixel: (premultiplied float[H][W][4], covered uint8[H][W])"""
    col = _f32(col)
    Z, Y, X, _ = col.shape
    out = np.empty((height, width, 4), np.float32)
    cov = np.empty((height, width), np.uint8)
    lmp = _fp(_f32(lightmap)) if lightmap is not None else None
    lib().orc_raycast_direct(_fp(col), lmp, X, Y, Z, C.byref(frame), _fp(_f32(wvp_i)), width, height, num_samples,
                             num_light_samples, int(has_sh), int(separate), _fp(out), cov.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out, cov


def visualize_color(col, width, height):
    """PSVisualizeColor per screen pixel of a 2-D grid: col (1, Y, X, 4) or (Y, X, 4) -> premultiplied float[H][W][4]"""
    col = _f32(col).reshape(col.shape[-3], col.shape[-2], 4)
    Y, X, _ = col.shape
    out = np.empty((height, width, 4), np.float32)
    lib().orc_visualize_color(_fp(col), X, Y, width, height, _fp(out))
    return out


def bc6h_decode_blocks(blocks):
    """blocks uint8[n][16] -> (half bit patterns uint16[n][16][3], modes int[n]) (mode 0 = reserved)"""
    blocks = np.ascontiguousarray(blocks, np.uint8).reshape(-1, 16)
    out = np.empty((len(blocks), 16, 3), np.uint16)
    modes = np.empty(len(blocks), np.int32)
    u8 = C.POINTER(C.c_uint8)
    for k in range(len(blocks)):
        modes[k] = lib().orc_bc6h_decode_block(blocks[k].ctypes.data_as(u8), out[k].ctypes.data_as(C.POINTER(C.c_uint16)))
    return out, modes


def dds_bc6h_cube(dds_bytes, mip=0):
    """DDS cube map (BC6H_UF16, DX10 header) -> (float32[6][n][n][3], mode histogram[15])"""
    buf = np.frombuffer(dds_bytes, np.uint8)
    u8 = C.POINTER(C.c_uint8)
    hist = (C.c_int * 15)()
    faces = []
    for f in range(6):
        tmp = np.zeros((4096 * 4096 * 3,), np.float32) if f == 0 else None
        if f == 0:
            n = lib().orc_dds_bc6h_cube_face(buf.ctypes.data_as(u8), len(buf), 0, mip, _fp(tmp), hist)
            if n <= 0:
                raise ValueError("not a BC6H_UF16 DDS cube map (or no such mip)")
            faces.append(tmp[:n * n * 3].reshape(n, n, 3).copy())
        else:
            a = np.empty((n, n, 3), np.float32)
            assert lib().orc_dds_bc6h_cube_face(buf.ctypes.data_as(u8), len(buf), f, mip, _fp(a), hist) == n
            faces.append(a)
    return np.stack(faces), np.array(list(hist))


def environment(cube, eye, s2w, width, height):
    """PSEnvironment per screen pixel: float cube [6][N][N][3] -> float[H][W][4] (rgb, alpha 0)"""
    cube = _f32(cube)
    out = np.empty((height, width, 4), np.float32)
    lib().orc_environment(_fp(cube), cube.shape[1], _fp(_f32(eye)), _fp(_f32(s2w)), width, height, _fp(out))
    return out


def world_view_proj_inverse(view, proj):
    """CBPerObject.WorldViewProjI as its four constant-buffer rows (Fluid.cpp:318)"""
    out = np.empty((4, 4), np.float32)
    lib().orc_world_view_proj_inverse(_fp(_f32(view)), _fp(_f32(proj)), _fp(out))
    return out


def resolve_cube(cube_u8, frame, wvp_i, width, height):
    """PSRayCastCube per screen pixel: (premultiplied float[H][W][4], covered uint8[H][W])"""
    cube_u8 = np.ascontiguousarray(cube_u8, np.uint8)
    N = cube_u8.shape[1]
    out = np.empty((height, width, 4), np.float32)
    cov = np.empty((height, width), np.uint8)
    u8 = C.POINTER(C.c_uint8)
    lib().orc_resolve_cube(cube_u8.ctypes.data_as(u8), N, C.byref(frame), _fp(_f32(wvp_i)), width, height, _fp(out), cov.ctypes.data_as(u8))
    return out, cov


def blend_premultiplied(src, covered, target_u8):
    """PREMULTIPLIED blend over an RGBA8 target; returns the new target"""
    t = np.ascontiguousarray(target_u8, np.uint8).copy()
    H, W = covered.shape
    u8 = C.POINTER(C.c_uint8)
    lib().orc_blend_premultiplied(_fp(_f32(src)), np.ascontiguousarray(covered, np.uint8).ctypes.data_as(u8), t.ctypes.data_as(u8), W, H)
    return t


def sh_transform(cube, quirk=False):
    cube = _f32(cube)
    N = cube.shape[1]
    out = np.empty((9, 3), np.float32)
    lib().orc_sh_transform(_fp(cube), N, _fp(out), int(quirk))
    return out


def sh_irradiance(sh, n):
    sh, n = _f32(sh), _f32(n)
    out = np.empty(3, np.float32)
    lib().orc_sh_irradiance(_fp(sh), _fp(n), _fp(out))
    return out
