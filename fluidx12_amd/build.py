"""Builds libfluidx_hip.so (the C-ABI library of include/fluidx_hip.h) in-tree with hipcc for gfx950.

    python -m fluidx12_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the
work-tree snapshot; `ensure_built()` rebuilds it there only when a source is newer than the library.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libfluidx_hip.so")
OBJDIR = os.path.join(HERE, "build")

SOURCES = ["fx_api.cpp", "fx_knobs.cpp", "fx_context.cpp", "fx_schedule.cpp", "fx_comm.cpp", "fx_checkpoint.cpp", "fx_sim.hip", "fx_advect_lds.hip", "fx_jacobi_strip.hip", "fx_jacobi_strip3.hip", "fx_jacobi_strip4.hip", "fx_jacobi_block.hip", "fx_jacobi_freeze.hip", "fx_jacobi_stripm.hip", "fx_jacobi2d.hip", "fx_render.hip", "fx_render_accel.hip", "fx_resolve.hip", "fx_bc6h.hip", "fx_sh.hip"]
HEADERS = ["fx_internal.h", "fx_context.h", "fx_host.h", "fx_hostmath.h", "fx_pk.h", "fx_march.h", "fx_knobs.h", os.path.join(ROOT, "include", "fluidx_hip.h")]

# -ffp-contract=off: the numerics contract allows a fused multiply-add only where the code
# says fmaf(); everything else is separately rounded, like the oracle.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-D__HIP_PLATFORM_AMD__"]
# FLUIDX_BUILD_LAB=1: the lab build -- every A/B switch of the launchers (fx_knobs.cpp) and the superseded kernels kept as their baselines
# (k_jacobi_strip4q ...).  The shipped library has ten switches and none of those kernels.
LAB = os.environ.get("FLUIDX_BUILD_LAB", "0") == "1"
if LAB:
    FLAGS = FLAGS + ["-DFX_LAB"]


# per-source additions.  The strip kernels run one wave per SIMD with ~300 registers: the machine scheduler's default
# (occupancy-driven) strategy has nothing to win there; "max-ilp" orders for instruction-level parallelism instead
# (k_jacobi_strip3 44.4 -> 43.6 us per launch; "max-memory-clause" 43.8, "iterative-ilp" 48.5).  Scheduling only: results are
# bit-identical (tests/test_gpu_sim.py).
EXTRA_FLAGS = {"fx_jacobi_strip3.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"] + (["-DFX_STRIP3C_ROWS=" + os.environ["FLUIDX_BUILD_STRIP3C_ROWS"]] if os.environ.get("FLUIDX_BUILD_STRIP3C_ROWS") else []),
               "fx_jacobi_strip.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
               "fx_jacobi_strip4.hip": (os.environ["FLUIDX_BUILD_STRIP4_SCHED"].split() if os.environ.get("FLUIDX_BUILD_STRIP4_SCHED") is not None else ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]) + (["-DFX_STRIP4_OUTER_ROWS=" + os.environ["FLUIDX_BUILD_STRIP4_OUTER_ROWS"]] if os.environ.get("FLUIDX_BUILD_STRIP4_OUTER_ROWS") else []) + os.environ.get("FLUIDX_BUILD_STRIP4_DEFS", "").split(),
               "fx_render_accel.hip": [("-D%s=%s" % (d, os.environ[e])) for e, d in (("FLUIDX_BUILD_MASK_BITS", "FX_MASK_BUDGET_BITS"), ("FLUIDX_BUILD_VIEW_AHEAD", "FX_VIEW_AHEAD"), ("FLUIDX_BUILD_LIGHT_AHEAD", "FX_LIGHT_AHEAD"), ("FLUIDX_BUILD_LIGHT_RAY_WGS", "FX_LIGHT_RAY_WGS"), ("FLUIDX_BUILD_LIGHT_RAY_UNROLL", "FX_LIGHT_RAY_UNROLL"), ("FLUIDX_BUILD_VIEW_NT", "FX_VIEW_NT"), ("FLUIDX_BUILD_VIEW_WPE", "FX_VIEW_WPE")) if os.environ.get(e)],
               # k_freeze_tiles reserves its list slot with a returning atomic whose round trip is meant to pass behind the staging loads;
               # the wave-aggregating atomic optimizer would wait for it on the spot (readfirstlane of the result)
               "fx_jacobi_freeze.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"] + (["-DFX_FREEZE_PROF"] if os.environ.get("FLUIDX_BUILD_FREEZE_PROF") else [])}     # 256x256x64: 8.9 -> 8.1 us per sweep; 512x512x64: 19.0 -> 18.8


def kernel_source_hash(kernel):
    """16 hex digits identifying the device code of `kernel` (a __global__ name such as k_jacobi_strip3c): sha256 over the
    .hip file that defines it, the device-side headers it can include and its compile flags.  profiles/*_pmc_traffic*.json and
    *_sq_counters*.json carry this stamp per kernel; bench.py refuses a counter summary whose stamp differs from the tree's."""
    import hashlib
    import re
    pat = re.compile(r"\bvoid\s+" + re.escape(kernel) + r"\s*\(")
    for s in SOURCES:
        if not s.endswith(".hip"):
            continue
        text = open(os.path.join(CSRC, s), "rb").read()
        if pat.search(text.decode("utf-8", "replace")):
            h = hashlib.sha256()
            h.update(text)
            h.update(open(os.path.join(CSRC, "fx_pk.h"), "rb").read())
            if b'#include "fx_march.h"' in text:                # the marches of the render kernels live in this header
                h.update(open(os.path.join(CSRC, "fx_march.h"), "rb").read())
            h.update(" ".join(FLAGS + EXTRA_FLAGS.get(s, [])).encode())
            return h.hexdigest()[:16]
    return None


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


def _newer(path, deps):
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    cc = hipcc()
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, s + ".o")
        objs.append(obj)
        flags = " ".join(FLAGS + EXTRA_FLAGS.get(s, []))
        stamp = obj + ".flags"                         # the flag string the object was built with: an object built under another
        built_with = open(stamp).read() if os.path.exists(stamp) else None   # FLUIDX_BUILD_* setting is stale whatever its mtime says
        if force or built_with != flags or _newer(obj, [src] + hdrs):
            cmd = [cc] + FLAGS + EXTRA_FLAGS.get(s, []) + ["-x", "hip", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            with open(stamp, "w") as fh:
                fh.write(flags)
    if force or _newer(LIB, objs):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


def build_variant(name, defs, sources=("fx_jacobi_strip4.hip",), outdir=None):
    """A lab build of the library: `sources` recompiled with the extra `defs` (e.g. ["-DFX_S4_NOHAND"]: leave-one-out timing builds of the
    four-sweep kernels, docs/LAB.md), every other object taken from the regular build; -> <outdir>/libfluidx_hip_<name>.so.  The tools
    load it with FLUIDX_LIB_PATH=<that file> (fluidx12_amd/capi.py).  Built HERE (hipcc cross-compiles) so that it travels to the GPU
    box with the snapshot; never what tests or bench.py measure."""
    build_lib()
    outdir = outdir or os.path.join(ROOT, "tools", "_variants")
    vdir = os.path.join(OBJDIR, "variant_" + name)
    os.makedirs(outdir, exist_ok=True)
    os.makedirs(vdir, exist_ok=True)
    objs = []
    for s in SOURCES:
        if s in sources:
            obj = os.path.join(vdir, s + ".o")
            subprocess.check_call([hipcc()] + FLAGS + EXTRA_FLAGS.get(s, []) + list(defs) + ["-x", "hip", "-c", os.path.join(CSRC, s), "-o", obj])
        else:
            obj = os.path.join(OBJDIR, s + ".o")
        objs.append(obj)
    lib = os.path.join(outdir, "libfluidx_hip_%s.so" % name)
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"])
    return lib


def ensure_built():
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    stale_flags = any(not os.path.exists(os.path.join(OBJDIR, s + ".o.flags")) or
                      open(os.path.join(OBJDIR, s + ".o.flags")).read() != " ".join(FLAGS + EXTRA_FLAGS.get(s, [])) for s in SOURCES)
    if _newer(LIB, srcs + hdrs) or (stale_flags and os.path.isdir(OBJDIR) and shutil.which("hipcc")):
        build_lib()
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
