"""fluidx12_amd -- MI355X-native smoke solver + cube-map-space ray marcher.

The product is libfluidx_hip.so (hand-written gfx950 HIP kernels behind the C ABI of
include/fluidx_hip.h).  This package holds its sources (csrc/), the in-tree build (build.py),
the ctypes declarations (capi.py) and the host-side mirror of the reference's `Fluid` /
`LightProbe` operators (fluid.py).  Nothing here imports oracle/.
"""
from .capi import (FluidxError, FIELD_VELOCITY, FIELD_VELOCITY1, FIELD_COLOR, FIELD_COLOR_PREV, FIELD_PRESSURE,
                   FIELD_DIVERGENCE, FIELD_LIGHTMAP, FIELD_CUBEMAP, FIELD_TARGET, FIELD_TARGET_FLOAT)
from .checkpoint import read_checkpoint, write_checkpoint
from .fluid import Fluid, LightProbe, comm_init_local, comm_init_peer, comm_unique_id, default_camera, look_at_lh, perspective_fov_lh

__all__ = ["Fluid", "LightProbe", "FluidxError", "comm_init_local", "comm_init_peer", "comm_unique_id", "default_camera",
           "look_at_lh", "perspective_fov_lh", "read_checkpoint", "write_checkpoint", "FIELD_VELOCITY", "FIELD_VELOCITY1", "FIELD_COLOR",
           "FIELD_COLOR_PREV", "FIELD_PRESSURE", "FIELD_DIVERGENCE", "FIELD_LIGHTMAP", "FIELD_CUBEMAP",
           "FIELD_TARGET", "FIELD_TARGET_FLOAT"]
