"""Host-side mirror of the reference's operator surface for the hot path, over the C ABI.

`Fluid` keeps the method names, argument meaning and error behaviour of
/root/reference/FluidX12/Content/Fluid.h:20-35 (Init returns bool, the rest return None and
raise only on ABI failure); `LightProbe` mirrors the SH side of Content/LightProbe.h:16-26.
The XUSG arguments that have no HIP meaning (descriptor-table lib, uploaders, RT/DS formats)
are dropped; `CommandList*` becomes an optional HIP stream handle (int / None).

This module only marshals: all arithmetic happens in libfluidx_hip.so.
"""
import ctypes as C
import os

import numpy as np

from . import capi


def look_at_lh(eye, focus, up):
    """XMMatrixLookAtLH (row-vector convention), as used at FluidX12/FluidX12.cpp:252."""
    eye, focus, up = (np.asarray(v, np.float32) for v in (eye, focus, up))
    z = focus - eye
    z = z / np.float32(np.sqrt(np.dot(z, z)))
    x = np.cross(up, z).astype(np.float32)
    x = x / np.float32(np.sqrt(np.dot(x, x)))
    y = np.cross(z, x).astype(np.float32)
    m = np.zeros((4, 4), np.float32)
    m[:3, 0], m[:3, 1], m[:3, 2] = x, y, z
    m[3, :3] = [-np.dot(x, eye), -np.dot(y, eye), -np.dot(z, eye)]
    m[3, 3] = 1.0
    return m


def perspective_fov_lh(fovy, aspect, zn, zf):
    """XMMatrixPerspectiveFovLH, as used at FluidX12/FluidX12.cpp:244."""
    h = np.float32(np.cos(0.5 * fovy) / np.sin(0.5 * fovy))
    w = np.float32(h / np.float32(aspect))
    q = np.float32(zf / (zf - zn))
    m = np.zeros((4, 4), np.float32)
    m[0, 0], m[1, 1], m[2, 2], m[2, 3], m[3, 2] = w, h, q, 1.0, -q * np.float32(zn)
    return m


def default_camera(width, height):
    """The demo driver's camera (FluidX12.cpp:243-253): eye (4,16,-40) -> origin, FOV pi/4, z 1..1000."""
    eye = np.array([4.0, 16.0, -40.0], np.float32)
    view = look_at_lh(eye, [0, 0, 0], [0, 1, 0])
    proj = perspective_fov_lh(np.float32(np.pi) / np.float32(4.0), width / float(height), 1.0, 1000.0)
    return view, proj, eye


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Fluid:
    """class Fluid (Content/Fluid.h:9-128) on HIP."""

    RAY_MARCH_DIRECT = capi.RAY_MARCH_DIRECT
    RAY_MARCH_CUBEMAP = capi.RAY_MARCH_CUBEMAP
    SEPARATE_LIGHT_PASS = capi.SEPARATE_LIGHT_PASS
    OPTIMIZED = capi.OPTIMIZED
    FrameCount = capi.FRAME_COUNT

    def __init__(self):
        self._lib = capi.load()
        self._ctx = C.c_void_p()
        self.grid = None
        self.slab = None
        self.last_status = capi.FX_OK

    # ---- Fluid::Init (Fluid.cpp:189-270) ----------------------------------------------------------
    def Init(self, width, height, gridSize, *, storage="fp32", jacobi_iters=40, jacobi_mode="fixed",
             advect_address="clamp", device=-1, slab=None, halo_advect=0, halo_jacobi=0, jacobi_fuse=0, overlap=2,
             render_only=False):
        if self._ctx:
            self.Release()
        X, Y, Z = (int(v) for v in gridSize)
        d = capi.Desc()
        d.struct_size = C.sizeof(capi.Desc)
        d.grid_x, d.grid_y, d.grid_z = X, Y, Z
        d.viewport_w, d.viewport_h = int(width), int(height)
        d.storage = {"fp32": capi.STORAGE_FP32, "fp16": capi.STORAGE_FP16}[storage]
        d.jacobi_iters = int(jacobi_iters)
        d.jacobi_mode = {"fixed": capi.JACOBI_FIXED, "faithful": capi.JACOBI_FAITHFUL}[jacobi_mode]
        d.advect_address = {"clamp": capi.ADDRESS_CLAMP, "mirror": capi.ADDRESS_MIRROR}[advect_address]
        d.device = int(device)
        if slab is not None:
            d.slab_z0, d.slab_nz = int(slab[0]), int(slab[1])
        d.halo_advect, d.halo_jacobi = int(halo_advect), int(halo_jacobi)
        level = 2 if overlap is True else int(overlap)        # 0 none, 1 advection halo only, 2 (default) + pressure rounds, 3 + early colour halo
        d.flags = (int(jacobi_fuse) & 0xF) | (0 if level else capi.FLAG_NO_OVERLAP) | (capi.FLAG_RENDER_ONLY if render_only else 0)
        self.last_status = self._lib.fx_create(C.byref(self._ctx), C.byref(d))
        if self.last_status != capi.FX_OK:      # the reference's Init returns false (XUSG_N_RETURN)
            self._ctx = C.c_void_p()
            return False
        self.grid = (X, Y, Z)
        self.viewport = (int(width), int(height))
        self.slab = (d.slab_z0, d.slab_nz if d.slab_nz else Z)
        if level in (1, 3):
            self.set_option(capi.OPT_OVERLAP, level)
        return True

    def Release(self):
        if self._ctx:
            self._lib.fx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.Release()
        except Exception:
            pass

    def _need(self):
        if not self._ctx:
            raise capi.FluidxError(capi.FX_E_STATE, "Fluid (Init not called)")

    # ---- Fluid.h:28-33 ----------------------------------------------------------------------------
    def SetMaxSamples(self, maxRaySamples, maxLightSamples):
        self._need()
        capi.check(self._lib.fx_set_max_samples(self._ctx, maxRaySamples, maxLightSamples), "SetMaxSamples")

    def SetSH(self, coeffSH):
        self._need()
        if coeffSH is None:
            capi.check(self._lib.fx_set_sh(self._ctx, None), "SetSH")
            return
        c = np.ascontiguousarray(coeffSH, np.float32).reshape(27)
        capi.check(self._lib.fx_set_sh(self._ctx, _fp(c)), "SetSH")

    def UpdateFrame(self, timeStep, frameIndex, view=None, proj=None, eyePt=None):
        self._need()
        if view is None:
            capi.check(self._lib.fx_update_frame(self._ctx, timeStep, frameIndex, None, None, None), "UpdateFrame")
            return
        v = np.ascontiguousarray(view, np.float32).reshape(16)
        p = np.ascontiguousarray(proj, np.float32).reshape(16)
        e = np.ascontiguousarray(eyePt, np.float32).reshape(3)
        capi.check(self._lib.fx_update_frame(self._ctx, timeStep, frameIndex, _fp(v), _fp(p), _fp(e)), "UpdateFrame")

    def Simulate(self, frameIndex=0, stream=None):
        self._need()
        capi.check(self._lib.fx_simulate(self._ctx, stream, frameIndex), "Simulate")

    def Render(self, frameIndex=0, flags=capi.OPTIMIZED, stream=None, to_target=False):
        """Fluid::Render (Fluid.cpp:412-446).  to_target=True also runs renderCube (Fluid.cpp:430), which the reference
        always does: the cube map is resolved onto the render target (ClearRenderTarget first, as the caller does)."""
        self._need()
        capi.check(self._lib.fx_render(self._ctx, stream, frameIndex, flags), "Render")
        if to_target and (flags & capi.RAY_MARCH_CUBEMAP):     # the direct modes write the target themselves
            self.RenderCube(frameIndex, stream)

    def ClearRenderTarget(self, rgba=(0.2, 0.2, 0.2, 0.0), stream=None):
        """ClearRenderTargetView with the demo's clear colour (FluidX12.cpp:471-472)"""
        self._need()
        c = (C.c_float * 4)(*[float(v) for v in rgba])
        capi.check(self._lib.fx_clear_render_target(self._ctx, stream, c), "ClearRenderTarget")

    def SetEnvironment(self, radiance_cube):
        """the radiance cube float[6][n][n][3] the sky pass draws (None releases it)"""
        self._need()
        if radiance_cube is None:
            capi.check(self._lib.fx_set_environment(self._ctx, None, 0), "SetEnvironment")
            return
        a = np.ascontiguousarray(radiance_cube, np.float32)
        if a.ndim != 4 or a.shape[0] != 6 or a.shape[1] != a.shape[2] or a.shape[3] != 3:
            raise ValueError("radiance cube must be float[6][n][n][3]")
        capi.check(self._lib.fx_set_environment(self._ctx, _fp(a), a.shape[1]), "SetEnvironment")

    def RenderEnvironment(self, frameIndex=0, stream=None):
        """LightProbe::RenderEnvironment (LightProbe.cpp:85-97): the sky onto the render target, before the volume"""
        self._need()
        capi.check(self._lib.fx_render_environment(self._ctx, stream, frameIndex), "RenderEnvironment")

    def RenderCube(self, frameIndex=0, stream=None):
        """Fluid::renderCube (Fluid.cpp:910-931), raster-free: PSRayCastCube per pixel + PREMULTIPLIED blend"""
        self._need()
        capi.check(self._lib.fx_render_cube(self._ctx, stream, frameIndex), "RenderCube")

    # ---- the demo driver's time-step rule (FluidX12.cpp:266) ---------------------------------------
    def default_time_step(self):
        X, Y, Z = self.grid
        return (2.0 if Z > 1 else 1.0) / Y

    # ---- stages / access (no reference counterpart) ---------------------------------------------------
    def Synchronize(self):
        self._need()
        rc = self._lib.fx_synchronize(self._ctx)
        if rc != capi.FX_OK:                       # what the library had to say beyond the status (fx_last_error)
            note = self.last_error
            capi.check(rc, "Synchronize" + (" [%s]" % note if note else ""))

    @property
    def last_error(self):
        self._need()
        e = self._lib.fx_last_error(self._ctx)
        return e.decode() if e else ""

    def Advect(self, stream=None):
        capi.check(self._lib.fx_advect(self._ctx, stream), "Advect")

    def Divergence(self, stream=None):
        capi.check(self._lib.fx_divergence(self._ctx, stream), "Divergence")

    def Jacobi(self, iters, stream=None):
        capi.check(self._lib.fx_jacobi(self._ctx, stream, iters), "Jacobi")

    def Project(self, stream=None):
        capi.check(self._lib.fx_project(self._ctx, stream), "Project")

    def frame_info(self):
        self._need()
        fi = capi.FrameInfo()
        capi.check(self._lib.fx_get_frame_info(self._ctx, C.byref(fi)), "frame_info")
        return fi

    def _shape(self, field):
        X, Y, _ = self.grid
        nz = self.slab[1]
        if field in (capi.FIELD_VELOCITY, capi.FIELD_VELOCITY1):
            return (3, nz, Y, X), np.float32
        if field in (capi.FIELD_COLOR, capi.FIELD_COLOR_PREV):
            return (nz, Y, X, 4), np.float32
        if field in (capi.FIELD_PRESSURE, capi.FIELD_DIVERGENCE):
            return (nz, Y, X), np.float32
        if field == capi.FIELD_LIGHTMAP:
            return (nz, Y, X, 3), np.float32
        if field == capi.FIELD_TARGET:
            return (self.viewport[1], self.viewport[0], 4), np.uint8
        if field == capi.FIELD_TARGET_FLOAT:
            return (self.viewport[1], self.viewport[0], 4), np.float32
        s = self.frame_info().cube_size
        return (6, s, s, 4), np.uint8

    def download(self, field):
        self._need()
        shape, dt = self._shape(field)
        out = np.empty(shape, dt)
        capi.check(self._lib.fx_download(self._ctx, field, out.ctypes.data_as(C.c_void_p), out.nbytes), "download")
        return out

    def digest(self, field, z_begin=0, z_count=0):
        """128-bit device-side digest (an int) of global planes [z_begin, z_begin + z_count) of a simulation field -- equal between a
        slab context and a single-domain context iff the planes agree bit for bit (fx_field_digest); z_count = 0: all owned planes"""
        self._need()
        out = (C.c_uint64 * 2)()
        capi.check(self._lib.fx_field_digest(self._ctx, field, int(z_begin), int(z_count), out), "digest")
        return (int(out[1]) << 64) | int(out[0])

    def upload(self, field, array):
        self._need()
        shape, dt = self._shape(field)
        a = np.ascontiguousarray(array, dt)
        if a.shape != shape:
            raise ValueError("field %d expects shape %s, got %s" % (field, shape, a.shape))
        capi.check(self._lib.fx_upload(self._ctx, field, a.ctypes.data_as(C.c_void_p), a.nbytes), "upload")

    def SaveCheckpoint(self, path):
        """velocity[0], colour[parity] and pressure of this context's planes into the whole-grid file `path` (every slab
        context of a chain saves to the same path); resuming from it continues bit-identically"""
        self._need()
        capi.check(self._lib.fx_checkpoint_save(self._ctx, os.fsencode(path)), "SaveCheckpoint")

    def LoadCheckpoint(self, path):
        self._need()
        capi.check(self._lib.fx_checkpoint_load(self._ctx, os.fsencode(path)), "LoadCheckpoint")

    def timing_enable(self, on=True):
        capi.check(self._lib.fx_timing_enable(self._ctx, int(on)), "timing_enable")

    def timing_read(self, reset=True):
        t = capi.Timing()
        capi.check(self._lib.fx_timing_read(self._ctx, C.byref(t), int(reset)), "timing_read")
        return t

    def set_option(self, option, value):
        """slab schedule knobs (capi.OPT_OVERLAP 0..3, capi.OPT_JACOBI_ROUND 1..halo_jacobi); same on every rank"""
        self._need()
        capi.check(self._lib.fx_set_option(self._ctx, int(option), int(value)), "set_option")

    def gather_color(self, full=None, root=0, slabs=None, stream=None):
        """multi-GPU rendering, exact: every rank's colour planes travel to `root`'s whole-grid context `full` (pass it on
        the root, None elsewhere); slabs = [(z0, nz)] * nranks for RCCL groups (a loop-back group knows its members)."""
        self._need()
        z0 = nz = None
        if slabs is not None:
            z0 = (C.c_uint32 * len(slabs))(*[int(a) for a, _ in slabs])
            nz = (C.c_uint32 * len(slabs))(*[int(b) for _, b in slabs])
        capi.check(self._lib.fx_comm_gather_color(self._ctx, stream, full._ctx if full is not None else None, int(root), z0, nz),
                   "gather_color")

    # ---- multi-GPU slabs ---------------------------------------------------------------------------------
    def comm_init_rank(self, unique_id, rank, nranks):
        buf = (C.c_char * len(unique_id)).from_buffer_copy(unique_id)
        capi.check(self._lib.fx_comm_init_rank(self._ctx, buf, len(unique_id), rank, nranks), "comm_init_rank")


def comm_unique_id():
    lib = capi.load()
    n = lib.fx_comm_id_bytes()
    buf = (C.c_char * n)()
    capi.check(lib.fx_comm_get_unique_id(buf, n), "comm_get_unique_id")
    return bytes(buf)


#: what `comm_init_local` builds when the caller does not say: the shared-stream group, or (tests re-running the slab suite with
#: real concurrency) the peer group
default_local_group = "shared"


def comm_init_local(fluids, peer=None):
    """In-process slab group of several `Fluid` slab contexts, driven through the first.  peer=False: one device, every member on one
    compute stream (fx_comm_init_local); peer=True: every member on its own streams and, if created so, its own device -- halo planes
    are pulled out of the neighbour's memory (fx_comm_init_peer)."""
    lib = capi.load()
    arr = (C.c_void_p * len(fluids))(*[f._ctx for f in fluids])
    if peer is None:
        peer = default_local_group == "peer"
    if peer:
        capi.check(lib.fx_comm_init_peer(arr, len(fluids)), "comm_init_peer")
    else:
        capi.check(lib.fx_comm_init_local(arr, len(fluids)), "comm_init_local")


def comm_init_peer(fluids):
    comm_init_local(fluids, peer=True)


class LightProbe:
    """SH side of class LightProbe (Content/LightProbe.h:16-26): TransformSH + GetSH."""

    def __init__(self, fluid):
        self._fluid = fluid
        self._radiance = None
        self._sh = None

    def Init(self, radiance_cube, mip=0):
        """radiance: a float cube [6][N][N][3], or -- like the reference (LightProbe.cpp:41-46) -- a DDS cube map in
        BC6H_UF16 given as a file name or its bytes (mip `mip` is decoded on the device)."""
        if isinstance(radiance_cube, (str, bytes, bytearray)):
            data = open(radiance_cube, "rb").read() if isinstance(radiance_cube, str) else bytes(radiance_cube)
            radiance_cube = self.decode_dds(data, mip)
            if radiance_cube is None:
                return False
        a = np.ascontiguousarray(radiance_cube, np.float32)
        if a.ndim != 4 or a.shape[0] != 6 or a.shape[1] != a.shape[2] or a.shape[3] != 3:
            return False
        self._radiance = a
        return True

    def decode_dds(self, data, mip=0):
        """DDS (BC6H_UF16 cube) bytes -> float32[6][n][n][3], or None if the container is something else"""
        f = self._fluid
        f._need()
        size, mips = C.c_uint32(), C.c_uint32()
        buf = (C.c_char * len(data)).from_buffer_copy(data)
        if f._lib.fx_dds_cube_info(buf, len(data), C.byref(size), C.byref(mips)) != capi.FX_OK or mip >= mips.value:
            return None
        n = max(size.value >> mip, 1)
        out = np.empty((6, n, n, 3), np.float32)
        capi.check(f._lib.fx_dds_decode_cube(f._ctx, buf, len(data), mip, _fp(out), out.size), "decode_dds")
        return out

    def TransformSH(self):
        f = self._fluid
        f._need()
        out = np.empty((9, 3), np.float32)
        capi.check(f._lib.fx_sh_transform(f._ctx, _fp(self._radiance), self._radiance.shape[1], _fp(out)), "TransformSH")
        self._sh = out

    def GetSH(self):
        return self._sh
