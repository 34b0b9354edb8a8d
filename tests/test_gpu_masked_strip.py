"""The masked strip kernels on their own: `fx::launch_freeze_strip4` (k_freeze_strip4o, fx_jacobi_strip4.hip) and `fx::launch_freeze_strip3`
(k_freeze_strip3, fx_jacobi_stripm.hip) called directly -- the C++ launchers through their mangled names, device memory from torch --
against a numpy model of the reference's loop (CSPoisson.hlsli:8-26) started from RANDOM frozen cells.  Inside a solve the frozen set
is smooth (a blob that shrinks), and an x-neighbour taken from the wrong cell of the adjacent lane hid behind "7 % of the cells differ";
random flags next to relaxing cells show it in the first quad.  Checked: both pressure copies, both mask copies, the tile marks and the
statistics word, bit for bit."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

f32 = np.float32
INV = np.uint32(0x3E2AAAAB).view(f32)          # 1/6 as the shader's constant
BELOW = f32(0.00100000005)                     # CSPoisson.hlsli:24 as compiled


class Geom(ctypes.Structure):                  # fx_internal.h struct Geom
    _fields_ = [(n, ctypes.c_int) for n in ("X", "Y", "Zg", "z0", "nz", "H", "zlo", "zhi")]


def launcher(levels):
    from fluidx12_amd import build
    build.ensure_built()
    out = subprocess.run(["nm", "-D", "--defined-only", build.LIB], capture_output=True, text=True, check=True).stdout
    names = re.findall(r"\b(_ZN2fx20launch_freeze_strip%dE\w+)" % levels, out)
    assert len(names) == 1, names
    fn = getattr(ctypes.CDLL(build.LIB), names[0])
    fn.restype = ctypes.c_int
    return fn


def clamped(a, axis, d):
    r = np.roll(a, d, axis)
    idx = [slice(None)] * 3
    idx[axis] = 0 if d == 1 else -1
    r[tuple(idx)] = a[tuple(idx)]
    return r


def masked_sweep(p, b, frozen):
    """one level of the loop for every cell: the sum in the shader's association order, the change by one fused multiply-add"""
    s = (((((clamped(p, 2, 1) - b) + clamped(p, 2, -1)) + clamped(p, 1, 1)) + clamped(p, 1, -1)) + clamped(p, 0, 1)) + clamped(p, 0, -1)
    x = (s * INV).astype(f32)
    change = (s.astype(np.float64) * np.float64(INV) - p.astype(np.float64)).astype(f32)       # fma(s, 1/6, -p): exact product, one rounding
    return np.where(frozen, p, x).astype(f32), frozen | (np.abs(change) < BELOW)


def nibbles(frozen):
    Z, Y, X = frozen.shape
    out = np.zeros((Z, Y, X // 4), np.uint8)
    for i in range(4):
        out |= frozen[:, :, i::4].astype(np.uint8) << i
    return out


@pytest.mark.parametrize("levels", [4, 3])
@pytest.mark.parametrize("depth,frac,amp,flat", [(8, 0.3, 1.0, False), (9, 0.3, 1.0, False), (16, 0.0, 1000.0, True), (16, 0.3, 1000.0, True), (16, 1.0, 1.0, True),
                                                 (16, 0.0, 1.0, False), (27, 0.3, 1.0, False), (64, 0.5, 1.0, False)])
def test_masked_strip_kernel_against_a_numpy_model(levels, depth, frac, amp, flat, rows=256):
    """depth: chunks of unequal length, the pipeline's fill and drain at both faces; frac: the share of cells frozen on entry (0: the loop
    freezes them itself, 1: the input comes back); amp = 1000 on a flat field: nothing new ever freezes (four plain sweeps around the
    frozen cells); the blob: most cells freeze in the first level"""
    import torch
    X, Y = 256, rows
    Z = depth
    rng = np.random.default_rng(100 * depth + levels)
    zz, yy, xx = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
    env = np.ones((Z, Y, X)) if flat else np.exp(-(((xx - 120) / 40.0) ** 2 + ((yy - 100) / 50.0) ** 2 + ((zz - Z / 2) / (Z / 3.0)) ** 2))
    p = (rng.standard_normal((Z, Y, X)) * 0.02 * env * amp).astype(f32)
    b = (rng.standard_normal((Z, Y, X)) * 0.01 * env * amp).astype(f32)
    frozen = rng.random((Z, Y, X)) < frac
    want_p, want_f = p.copy(), frozen.copy()
    relaxing_after = [bool((~want_f).any())]
    for _ in range(levels):
        want_p, want_f = masked_sweep(want_p, b, want_f)
        relaxing_after.append(bool((~want_f).any()))
    dev = "cuda"
    tp, tb, tm = torch.from_numpy(p).to(dev), torch.from_numpy(b).to(dev), torch.from_numpy(nibbles(frozen)).to(dev)
    tA, tB = torch.full_like(tp, 7.0), torch.full_like(tp, 9.0)
    tmA, tmB = torch.full_like(tm, 0x55), torch.full_like(tm, 0x66)
    marks = torch.zeros(((Z + 7) // 8) * (Y // 8) * 8, dtype=torch.int32, device=dev)
    stat = torch.zeros(4, dtype=torch.int32, device=dev)
    g = Geom(X, Y, Z, 0, Z, 0, 0, Z - 1)
    vp = ctypes.c_void_p
    TAG, LEVEL_IN = 77, 10
    rc = launcher(levels)(ctypes.byref(g), vp(tp.data_ptr()), vp(tb.data_ptr()), vp(tA.data_ptr()), vp(tB.data_ptr()), vp(tm.data_ptr()), vp(tmA.data_ptr()), vp(tmB.data_ptr()),
                          vp(marks.data_ptr()), ctypes.c_uint(TAG), vp(stat.data_ptr()), ctypes.c_uint(0), ctypes.c_int(LEVEL_IN), vp(0))
    assert rc == 0
    torch.cuda.synchronize()
    assert np.array_equal(tA.cpu().numpy().view(np.uint32), want_p.view(np.uint32))
    assert np.array_equal(tB.cpu().numpy().view(np.uint32), want_p.view(np.uint32))
    assert np.array_equal(tmA.cpu().numpy(), nibbles(want_f)) and np.array_equal(tmB.cpu().numpy(), nibbles(want_f))
    # a 32 x 8 x 8 tile with a cell that still relaxes carries the tag
    got_marks = marks.cpu().numpy().reshape(-1, Y // 8, 8)
    rel = np.zeros((got_marks.shape[0] * 8, Y, X), bool)
    rel[:Z] = ~want_f
    want_marks = rel.reshape(-1, 8, Y // 8, 8, 8, 32).any(axis=(1, 3, 5)) * TAG
    assert np.array_equal(got_marks, want_marks)
    # the statistics word: the last level that left a cell relaxing (LEVEL_IN itself: a cell that came in relaxing); untouched if none did
    last = max([l for l, r in enumerate(relaxing_after) if r], default=-1)
    assert int(stat.cpu().numpy()[0]) == (LEVEL_IN + last if last >= 0 else 0)


@pytest.mark.parametrize("rows", [128, 240, 24, 32, 72])
@pytest.mark.parametrize("depth,frac,amp,flat", [(9, 0.3, 1.0, False), (27, 0.0, 1.0, False), (16, 0.3, 1000.0, True)])
def test_masked_octet_on_rows_that_do_not_tile_its_bands(rows, depth, frac, amp, flat):
    """Y % 14 = 1 or 2 (128, 240, 72 ...): the second-to-last band's recomputed halo would reach beyond the last row, so that band is shifted
    up as well (octet_band_y, fx_jacobi_strip4.hip) -- round 5 computed level-l rows behind the wall from clamped loads there and two
    workgroups stored different bits to one address.  Every row count, against the numpy model bit for bit (pressures, nibbles, tile marks)"""
    test_masked_strip_kernel_against_a_numpy_model(4, depth, frac, amp, flat, rows=rows)


SOAK = int(os.environ.get("FLUIDX_MASKED_SOAK", "6"))       # random cases per kernel (a soak run: FLUIDX_MASKED_SOAK=200)


@pytest.mark.parametrize("levels", [4, 3])
@pytest.mark.parametrize("case", range(SOAK))
def test_masked_strip_kernel_on_random_cases(levels, case):
    rng = np.random.default_rng(7000 + case)
    depth = int(rng.integers(8, 70))
    frac = float(rng.choice([0.0, 0.05, 0.3, 0.6, 0.95]))
    amp = float(rng.choice([0.3, 1.0, 3.0, 1000.0]))
    test_masked_strip_kernel_against_a_numpy_model(levels, depth, frac, amp, bool(rng.integers(0, 2)))
