"""ctypes declarations of the C ABI in include/fluidx_hip.h (libfluidx_hip.so).

The library is the product: there is no Python/CPU fallback.  `load()` raises if the HIP
extension cannot be built or loaded.
"""
import ctypes as C
import os

from . import build as _build

FX_OK = 0
FX_E_INVALID, FX_E_DEVICE, FX_E_NOMEM, FX_E_STATE, FX_E_COMM, FX_E_HALO = -1, -2, -3, -4, -5, -6

RAY_MARCH_DIRECT, RAY_MARCH_CUBEMAP, SEPARATE_LIGHT_PASS, OPTIMIZED = 0, 1, 2, 3
FRAME_COUNT = 3
STORAGE_FP32, STORAGE_FP16 = 0, 1
JACOBI_FIXED, JACOBI_FAITHFUL = 0, 1
ADDRESS_CLAMP, ADDRESS_MIRROR = 0, 1
FLAG_JACOBI_FUSE_MASK, FLAG_NO_OVERLAP, FLAG_RENDER_ONLY = 0xF, 0x10, 0x20
OPT_OVERLAP, OPT_JACOBI_ROUND, OPT_ADAPTIVE_HALO, OPT_COUNT_SAMPLES, OPT_RENDER_ACCEL = 1, 2, 3, 4, 5
ABI_VERSION = 7                      # FX_ABI_VERSION of include/fluidx_hip.h
(FIELD_VELOCITY, FIELD_VELOCITY1, FIELD_COLOR, FIELD_COLOR_PREV, FIELD_PRESSURE, FIELD_DIVERGENCE,
 FIELD_LIGHTMAP, FIELD_CUBEMAP, FIELD_TARGET, FIELD_TARGET_FLOAT) = range(10)


class Desc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("grid_x", C.c_uint32), ("grid_y", C.c_uint32), ("grid_z", C.c_uint32),
                ("viewport_w", C.c_uint32), ("viewport_h", C.c_uint32), ("storage", C.c_uint32),
                ("jacobi_iters", C.c_uint32), ("jacobi_mode", C.c_uint32), ("advect_address", C.c_uint32),
                ("device", C.c_int32), ("slab_z0", C.c_uint32), ("slab_nz", C.c_uint32),
                ("halo_advect", C.c_uint32), ("halo_jacobi", C.c_uint32), ("flags", C.c_uint32)]


class FrameInfo(C.Structure):
    _fields_ = [("cube_lod", C.c_uint32), ("cube_size", C.c_uint32), ("ray_samples", C.c_uint32),
                ("visibility_mask", C.c_uint32), ("frame_parity", C.c_uint32), ("edge_pixels", C.c_float),
                ("time_step", C.c_float), ("world_view_proj_i", C.c_float * 16), ("screen_to_world", C.c_float * 16)]


class Timing(C.Structure):
    _fields_ = [("advect_ms", C.c_double), ("divergence_ms", C.c_double), ("jacobi_ms", C.c_double),
                ("project_ms", C.c_double), ("light_ms", C.c_double), ("view_ms", C.c_double),
                ("exchange_ms", C.c_double), ("steps", C.c_uint64), ("jacobi_launches", C.c_uint64),
                ("jacobi_sweeps", C.c_uint64), ("renders", C.c_uint64), ("resolve_ms", C.c_double),
                ("jacobi_main_ms", C.c_double), ("jacobi_main_launches", C.c_uint64), ("jacobi_main_sweeps", C.c_uint64),
                ("exchange_bytes", C.c_uint64), ("advect_halo_planes", C.c_uint64), ("chain_ms", C.c_double),
                ("freeze_solves", C.c_uint64), ("freeze_sweeps", C.c_uint64), ("exchange_calls", C.c_uint64),
                ("view_samples", C.c_uint64), ("light_samples", C.c_uint64), ("lightmap_fetches", C.c_uint64),
                ("freeze_strip_launches", C.c_uint64)]


# every symbol include/fluidx_hip.h declares: name -> (restype, argtypes)
_vp, _fp = C.c_void_p, C.POINTER(C.c_float)
SYMBOLS = {
    "fx_abi_version": (C.c_int, []),
    "fx_error_string": (C.c_char_p, [C.c_int]),
    "fx_create": (C.c_int, [C.POINTER(_vp), C.POINTER(Desc)]),
    "fx_destroy": (C.c_int, [_vp]),
    "fx_set_max_samples": (C.c_int, [_vp, C.c_uint32, C.c_uint32]),
    "fx_set_sh": (C.c_int, [_vp, _fp]),
    "fx_update_frame": (C.c_int, [_vp, C.c_float, C.c_uint8, _fp, _fp, _fp]),
    "fx_simulate": (C.c_int, [_vp, _vp, C.c_uint8]),
    "fx_render": (C.c_int, [_vp, _vp, C.c_uint8, C.c_uint8]),
    "fx_get_frame_info": (C.c_int, [_vp, C.POINTER(FrameInfo)]),
    "fx_synchronize": (C.c_int, [_vp]),
    "fx_last_error": (C.c_char_p, [_vp]),
    "fx_upload": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t]),
    "fx_download": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t]),
    "fx_field_bytes": (C.c_size_t, [_vp, C.c_int]),
    "fx_field_digest": (C.c_int, [_vp, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
    "fx_checkpoint_save": (C.c_int, [_vp, C.c_char_p]),
    "fx_checkpoint_load": (C.c_int, [_vp, C.c_char_p]),
    "fx_advect": (C.c_int, [_vp, _vp]),
    "fx_divergence": (C.c_int, [_vp, _vp]),
    "fx_jacobi": (C.c_int, [_vp, _vp, C.c_uint32]),
    "fx_project": (C.c_int, [_vp, _vp]),
    "fx_sh_transform": (C.c_int, [_vp, _fp, C.c_uint32, _fp]),
    "fx_set_environment": (C.c_int, [_vp, _fp, C.c_uint32]),
    "fx_render_environment": (C.c_int, [_vp, _vp, C.c_uint8]),
    "fx_clear_render_target": (C.c_int, [_vp, _vp, _fp]),
    "fx_render_cube": (C.c_int, [_vp, _vp, C.c_uint8]),
    "fx_dds_cube_info": (C.c_int, [_vp, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "fx_dds_decode_cube": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint32, _fp, C.c_size_t]),
    "fx_timing_enable": (C.c_int, [_vp, C.c_int]),
    "fx_set_option": (C.c_int, [_vp, C.c_uint32, C.c_uint32]),
    "fx_set_knob": (C.c_int, [C.c_char_p, C.c_char_p]),
    "fx_knob_name": (C.c_char_p, [C.c_uint32]),
    "fx_comm_gather_color": (C.c_int, [_vp, _vp, _vp, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "fx_timing_read": (C.c_int, [_vp, C.POINTER(Timing), C.c_int]),
    "fx_comm_id_bytes": (C.c_size_t, []),
    "fx_comm_get_unique_id": (C.c_int, [_vp, C.c_size_t]),
    "fx_comm_init_rank": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, C.c_int]),
    "fx_comm_init_local": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "fx_comm_init_peer": (C.c_int, [C.POINTER(_vp), C.c_int]),
}

_lib = None
# the launcher switches that exist in lab builds only (fx_knobs.cpp, -DFX_LAB)
LAB_KNOBS = {"ADVECT_BLOCK", "ADVECT_FAST", "ADVECT_LDS_HALF", "ADVECT_TILE_ROWS", "ADVECT_ZCHUNK", "BLOCK_REMAP", "BLOCK_SHAPE", "DEBUG_NO_COPY",
             "FREEZE_DENSE_LEVELS", "FREEZE_DENSE_ONE", "FREEZE_FAST", "FREEZE_FUSE_DIV", "FREEZE_NT", "FREEZE_SHRINK", "FREEZE_T", "FREEZE_WGS",
             "JACOBI2D_TILE", "JACOBI_BLOCK", "JACOBI_BLOCKG", "LIGHT_FILL_DIRTY", "LIGHT_RAY_NT", "LIGHT_RAY_WGS", "PROJECT_V4", "ROW_VW",
             "STRIP3H_PAIRS", "STRIP3_COOP", "STRIP3_NO512", "STRIP3_OFF", "STRIP3_ZCHUNK", "STRIP4T", "STRIP4T_256", "STRIP4T_512", "STRIP4T_FROM", "STRIP4T_GRID", "STRIP4T_NARROW", "STRIP4T_NARROW_FROM", "STRIP4T_PIECES", "STRIP4X", "STRIP4X_MINP", "STRIP4X_NT", "STRIP4X_ORDER", "STRIP4X_WGS",
             "STRIP4_OCTET", "STRIP4_ZCHUNK", "STRIP4_ZFLOOR", "STRIP_GENERIC", "STRIP_R", "STRIP_REMAP", "STRIP_WGS", "STRIP_WIDE", "STRIP_ZCHUNK", "VIEW_ORDER", "VIEW_WGS", "XCD_REMAP"}


def lib_path():
    return _build.LIB


def load():
    """Load libfluidx_hip.so (building it first if a source is newer).  Raises on failure."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # A process that also uses torch must share ONE HIP runtime: torch bundles its own
        # libamdhip64 and must be imported before this library resolves the same soname.
        import torch  # noqa: F401
    except Exception:  # torch is plumbing, not a requirement
        pass
    path = _build.ensure_built()
    if os.environ.get("FLUIDX_LIB_PATH"):          # a lab build of the library (build.build_variant): tools only -- tests and bench.py never set it
        path = os.environ["FLUIDX_LIB_PATH"]
    if not os.path.exists(path):
        raise RuntimeError("libfluidx_hip.so is missing and could not be built")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    # the version first: an older library lacks symbols, and "ABI version 5, this harness speaks 6" says more than an AttributeError
    lib.fx_abi_version.restype, lib.fx_abi_version.argtypes = C.c_int, []
    have = lib.fx_abi_version()
    if have != ABI_VERSION:
        raise RuntimeError("fluidx ABI version mismatch: %s speaks version %d, this harness version %d" % (path, have, ABI_VERSION))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    # the launcher switches (fx_set_knob) are process-wide values inside the library, which reads no environment for them; the tools'
    # habit of `FLUIDX_<NAME>=... python tools/...` is served here, once, by the harness
    offered = knob_names()
    for name in offered:
        v = os.environ.get("FLUIDX_" + name)
        if v is not None:
            set_knob(name, v)
    # ... and a tool that asks for a lab switch on the shipped library is told so instead of measuring the default twice
    import sys
    for k in os.environ:
        if k.startswith("FLUIDX_") and k[7:] in LAB_KNOBS and k[7:] not in offered:
            sys.stderr.write("fluidx12_amd: %s is a lab-build switch (FLUIDX_BUILD_LAB=1 python -m fluidx12_amd.build); this library ignores it\n" % k)
    return lib


def knob_names():
    lib, out, i = load(), [], 0
    while True:
        n = lib.fx_knob_name(i)
        if n is None:
            return out
        out.append(n.decode())
        i += 1


def set_knob(name, value):
    """a measurement switch of the kernel launchers (include/fluidx_hip.h fx_set_knob); value None = back to the default"""
    check(load().fx_set_knob(name.encode(), None if value is None else str(value).encode()), "set_knob(%s)" % name)


class FluidxError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = load().fx_error_string(status).decode()
        super().__init__("%s failed: %s (%d)" % (where, msg, status))


def check(status, where):
    if status != FX_OK:
        raise FluidxError(status, where)
