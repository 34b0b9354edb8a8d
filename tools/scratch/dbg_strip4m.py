# scratch: drives fx::launch_freeze_strip4 / strip3 directly (mangled C++ symbols) against a numpy emulation of the masked loop
import ctypes, sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fluidx12_amd import build
build.ensure_built()
L = ctypes.CDLL(build.LIB)
class Geom(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("X", "Y", "Zg", "z0", "nz", "H", "zlo", "zhi")]
f32 = np.float32
INV = np.uint32(0x3e2aaaab).view(f32)
KB = f32(0.00100000005)
def sweep(p, b, m):
    """one masked level: returns p', m' (bool frozen)"""
    def sh(a, ax, d):
        r = np.roll(a, d, ax)
        idx = [slice(None)] * 3
        if d == 1: idx[ax] = 0
        else: idx[ax] = -1
        r[tuple(idx)] = a[tuple(idx)]
        return r
    Lx, Rx = sh(p, 2, 1), sh(p, 2, -1)
    U, D = sh(p, 1, 1), sh(p, 1, -1)
    F, B = sh(p, 0, 1), sh(p, 0, -1)
    s = (((((Lx - b) + Rx) + U) + D) + F) + B
    x = (s * INV).astype(f32)
    t = (s.astype(np.float64) * np.float64(INV) - p.astype(np.float64)).astype(f32)
    fr = np.abs(t) < KB
    return np.where(m, p, x).astype(f32), m | fr
def run(Z, levels, seed=1, which=4, frac=0.3, amp=1.0, flat=False, time_it=False):
    X = Y = 256
    rng = np.random.default_rng(seed)
    zz, yy, xx = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
    env = np.exp(-(((xx - 120) / 40.0) ** 2 + ((yy - 100) / 50.0) ** 2 + ((zz - Z / 2) / (Z / 3.0)) ** 2))
    if flat: env = np.ones_like(env)
    p = (rng.standard_normal((Z, Y, X)) * 0.02 * env * amp).astype(f32)
    b = (rng.standard_normal((Z, Y, X)) * 0.01 * env * amp).astype(f32)
    m0 = rng.random((Z, Y, X)) < frac
    print("--- frac", frac, "amp", amp, "flat", flat)
    mb = np.zeros((Z, Y, X // 4), np.uint8)
    for i in range(4): mb |= (m0[:, :, i::4].astype(np.uint8) << i)
    pe, me = p.copy(), m0.copy()
    for _ in range(levels): pe, me = sweep(pe, b, me)
    mbe = np.zeros((Z, Y, X // 4), np.uint8)
    for i in range(4): mbe |= (me[:, :, i::4].astype(np.uint8) << i)
    dev = "cuda"
    tp, tb = torch.from_numpy(p).to(dev), torch.from_numpy(b).to(dev)
    tA, tB = torch.full_like(tp, 7.0), torch.full_like(tp, 9.0)
    tm = torch.from_numpy(mb).to(dev)
    tmA, tmB = torch.full_like(tm, 0x55), torch.full_like(tm, 0x66)
    marks = torch.zeros(((Z + 7) // 8) * 32 * 8, dtype=torch.int32, device=dev)
    stat = torch.zeros(4, dtype=torch.int32, device=dev)
    g = Geom(X, Y, Z, 0, Z, 0, 0, Z - 1)
    fn = getattr(L, "_ZN2fx20launch_freeze_strip%dERKNS_4GeomEPKfS4_PfS5_PKhPhS8_PjjS9_jiP12ihipStream_t" % which)
    fn.restype = ctypes.c_int
    vp = ctypes.c_void_p
    rc = fn(ctypes.byref(g), vp(tp.data_ptr()), vp(tb.data_ptr()), vp(tA.data_ptr()), vp(tB.data_ptr()), vp(tm.data_ptr()), vp(tmA.data_ptr()), vp(tmB.data_ptr()),
            vp(marks.data_ptr()), ctypes.c_uint(77), vp(stat.data_ptr()), ctypes.c_uint(0), ctypes.c_int(10), vp(0))
    torch.cuda.synchronize()
    if time_it:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn(ctypes.byref(g), vp(tp.data_ptr()), vp(tb.data_ptr()), vp(tA.data_ptr()), vp(tB.data_ptr()), vp(tm.data_ptr()), vp(tmA.data_ptr()), vp(tmB.data_ptr()),
               vp(marks.data_ptr()), ctypes.c_uint(77), vp(stat.data_ptr()), ctypes.c_uint(0), ctypes.c_int(10), vp(0))
        e1.record(); torch.cuda.synchronize()
        print("TIME which", which, "Z", Z, "%.1f us per launch" % (e0.elapsed_time(e1) * 1e3 / 50))
        return
    A, B, mA, mB = tA.cpu().numpy(), tB.cpu().numpy(), tmA.cpu().numpy(), tmB.cpu().numpy()
    mk = marks.cpu().numpy().reshape(-1, 32, 8)
    print("Z", Z, "which", which, "rc", rc, "pA ok", np.array_equal(A, pe), "pB ok", np.array_equal(B, pe), "mA ok", np.array_equal(mA, mbe), "mB ok", np.array_equal(mB, mbe), "stat", stat.cpu().numpy()[:1])
    # expected marks
    rel = ~me
    em = np.zeros_like(mk)
    for tz in range(mk.shape[0]):
        blk = rel[tz * 8:(tz + 1) * 8]
        em[tz] = blk.reshape(blk.shape[0], 32, 8, 8, 32).any(axis=(0, 2, 4)) * 77
    print("   marks ok", np.array_equal(mk, em), "n", (mk == 77).sum(), (em == 77).sum())
    for name, G, E in (("pA", A, pe), ("mA", mA, mbe)):
        d = np.argwhere(G != E)
        if len(d):
            print("  ", name, "ndiff", len(d), "z", np.unique(d[:, 0]), "y%14", np.bincount(d[:, 1] % 14, minlength=14), "first", d[:5].tolist())
            for c in d[:5]: print("     got", G[tuple(c)], "want", E[tuple(c)])

import os
if os.environ.get("DBG_TIME"):
    run(256, 4, frac=0.3, time_it=True)
    run(256, 3, which=3, frac=0.3, time_it=True)
else:
    run(16, 4, frac=0.0, amp=1000.0, flat=True)
    for Z in (8, 9, 16, 27, 64):
        run(Z, 4, frac=0.3)
