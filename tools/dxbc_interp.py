#!/usr/bin/env python3
"""Lock-step SIMT interpreter for the reference's shipped compute shaders (DXBC, cs_5_0).

Test infrastructure only.  `tools/dxbc.py` decodes /root/reference/Bin/*.cso; this module EXECUTES the decoded
program for every thread of a dispatch at once (numpy, one lane per thread, execution masks for structured
control flow) so that the reference's own binaries produce golden vectors (tools/make_dxbc_golden.py ->
tests/golden/dxbc_*.npz).  Nothing here is derived from the oracle or from the HIP kernels.

Execution model: ALL threads of the dispatch advance instruction by instruction ("lock step").  For the
in-place pressure relaxation of CSProject (globallycoherent UAV, no cross-group sync -> schedule dependent in
the reference) this is exactly the lock-step schedule the build defines: every load of a sweep happens before
any store of that sweep.

Arithmetic conventions (the D3D11 functional spec leaves these to the implementation; the same choices as
DESIGN.md's numerics contract, stated here independently):
  * fp32 IEEE add/mul/div/sqrt; `mad` is a fused multiply-add (exactly rounded, emulated in fp64 with a
    tie-breaking correction); `rsq` = 1/sqrt, `exp` = exp2, both correctly rounded from fp64;
  * min/max return the non-NaN operand; comparisons with NaN are false;
  * sample_l = trilinear filter with fp32 weights: t = u*N - 0.5, i0 = floor(t), f = t - i0, taps addressed by the
    sampler's mode (CLAMP / MIRROR / WRAP), blend = lerp x, then y, then z, lerp(a,b,f) = fma(f, b - a, a);
  * out-of-range typed loads return 0, out-of-range stores are dropped;
  * typed UAV stores convert to the resource format: R32_FLOAT, R16G16B16A16_FLOAT (RNE), R11G11B10_FLOAT
    (RNE, negatives -> 0), R8G8B8A8_UNORM (floor(255 sat(v) + 0.5)).
"""
import numpy as np

import dxbc

U32 = np.uint32
F32 = np.float32

# The OTHER choices the D3D11 functional spec allows an implementation (tools/dxbc_sensitivity.py switches them one at a time to bound
# what the pinned choices above are worth; the goldens are made with every switch at its default):
#   mad_fused      False: `mad` (and the multiply-add chain of dp2 / dp3 / dp4) rounds its product before the add
#   rsq_ulps       n != 0: `rsq` returns the correctly rounded result moved by n units in the last place (the spec allows 1 ulp)
#   exp_ulps       n != 0: the same for `exp`
#   filter_bits    n > 0: sample_l's blend weights carry n fractional bits (fixed point, the spec's minimum is 8) instead of fp32
ALT = {"mad_fused": True, "rsq_ulps": 0, "exp_ulps": 0, "filter_bits": 0}


def _ulps(r, n):
    r = r.astype(F32)
    for _ in range(abs(int(n))):
        r = np.nextafter(r, F32(np.inf) if n > 0 else F32(-np.inf)).astype(F32)
    return r


def mad32(a, b, c):
    """`mad` as the implementation under test computes it"""
    if ALT["mad_fused"]:
        return fma32(a, b, c)
    with np.errstate(all="ignore"):
        return ((a.astype(F32) * b.astype(F32)).astype(F32) + c.astype(F32)).astype(F32)


def f2u(a):
    return np.ascontiguousarray(a, dtype=F32).view(U32)


def u2f(a):
    return np.ascontiguousarray(a, dtype=U32).view(F32)


def fma32(a, b, c):
    """exactly rounded fp32 fma of fp32 arrays"""
    a64, b64, c64 = a.astype(np.float64), b.astype(np.float64), c.astype(np.float64)
    with np.errstate(all="ignore"):
        p = a64 * b64                      # exact: 24 x 24 bits
        s = p + c64                        # one rounding to 53 bits
        # error of that rounding (TwoSum)
        bb = s - p
        err = (p - (s - bb)) + (c64 - bb)
        r = s.astype(F32)
        # double rounding matters only if s sits exactly halfway between two floats and err != 0
        back = r.astype(np.float64)
        diff = s - back
        nxt = np.nextafter(r, np.where(diff > 0, F32(np.inf), F32(-np.inf))).astype(np.float64)
        half = (nxt - back) * 0.5
        tie = np.isfinite(s) & (err != 0) & (diff != 0) & (np.abs(diff) == np.abs(half))
        # at a tie the fp32 cast rounded to even; the true value lies on err's side of s
        toward_next = (np.sign(err) == np.sign(diff))
        fix = tie & toward_next
        r = np.where(fix, nxt.astype(F32), r)
        # if cast rounded *away* (to even = the farther one cannot happen: both equidistant) nothing else to do
    return r


def half_round(a):
    with np.errstate(over="ignore"):
        return a.astype(np.float16).astype(F32)


def pack_ufloat(f, mbits):
    x = f2u(f).astype(np.int64)
    out = np.zeros(x.shape, np.int64)
    drop = 23 - mbits
    maxfinite = (31 << mbits) - 1
    nan = (x & 0x7FFFFFFF) > 0x7F800000
    neg = (x & 0x80000000) != 0
    inf = x == 0x7F800000
    sub = x < 0x38800000
    with np.errstate(all="ignore"):
        q = np.rint(f.astype(np.float64) * 2.0 ** (14 + mbits)).astype(np.int64)
    v = x - (112 << 23)
    v = v + ((1 << (drop - 1)) - 1) + ((v >> drop) & 1)
    v = np.minimum(v >> drop, maxfinite)
    out = np.where(sub, q, v)
    out = np.where(inf, 31 << mbits, out)
    out = np.where(neg, 0, out)
    out = np.where(nan, (31 << mbits) | 1, out)
    return out


def unpack_ufloat(b, mbits):
    e, m = b >> mbits, b & ((1 << mbits) - 1)
    sub = m.astype(np.float64) * 2.0 ** (-14 - mbits)
    nor = (1.0 + m.astype(np.float64) * 2.0 ** (-mbits)) * 2.0 ** (e.astype(np.float64) - 15)
    out = np.where(e == 0, sub, nor)
    out = np.where(e == 31, np.where(m != 0, np.nan, np.inf), out)
    return out.astype(F32)


def quant_r11g11b10(rgb):
    out = np.empty_like(rgb)
    for c, mb in ((0, 6), (1, 6), (2, 5)):
        out[..., c] = unpack_ufloat(pack_ufloat(rgb[..., c], mb), mb)
    return out


def to_unorm8(v):
    with np.errstate(invalid="ignore"):
        s = np.where(v > 0, np.minimum(v, F32(1.0)), F32(0.0)).astype(F32)     # NaN -> 0
    return np.floor(s * F32(255.0) + F32(0.5)).astype(np.uint8)


class Texture:
    """typed resource [Z][Y][X][C] float32 (2D arrays: Z = slices)."""

    def __init__(self, data, fmt="R32G32B32A32_FLOAT"):
        self.data = np.ascontiguousarray(data, F32)
        assert self.data.ndim == 4
        self.fmt = fmt

    @property
    def dims(self):
        z, y, x, _ = self.data.shape
        return x, y, z

    def load(self, x, y, z, valid):
        X, Y, Z = self.dims
        ok = valid & (x >= 0) & (x < X) & (y >= 0) & (y < Y) & (z >= 0) & (z < Z)
        xs, ys, zs = np.clip(x, 0, X - 1), np.clip(y, 0, Y - 1), np.clip(z, 0, Z - 1)
        t = self.data[zs, ys, xs]
        C = t.shape[-1]
        out = np.zeros((len(x), 4), F32)
        out[:, :C] = t
        if C < 4:
            out[:, 3] = 1.0 if C == 3 else 0.0          # missing alpha reads 1, missing g/b read 0
        out[~ok] = 0.0
        return out

    def store(self, x, y, z, val, mask):
        X, Y, Z = self.dims
        ok = mask & (x >= 0) & (x < X) & (y >= 0) & (y < Y) & (z >= 0) & (z < Z)
        C = self.data.shape[-1]
        v = val[:, :C].astype(F32)
        if self.fmt == "R16G16B16A16_FLOAT":
            v = half_round(v)
        elif self.fmt == "R11G11B10_FLOAT":
            v = quant_r11g11b10(v)
        elif self.fmt == "R8G8B8A8_UNORM":
            v = to_unorm8(v).astype(F32) / F32(255.0)
        self.data[z[ok], y[ok], x[ok]] = v[ok]


class Structured:
    """structured buffer: [count][stride_dwords] uint32"""

    def __init__(self, count, stride_bytes, init=None):
        self.words = np.zeros((count, stride_bytes // 4), U32)
        if init is not None:
            self.words[...] = f2u(np.asarray(init, F32)).reshape(self.words.shape)


class Sampler:
    def __init__(self, address="CLAMP"):
        self.address = address

    def tap(self, i, n):
        if self.address == "MIRROR":
            m = np.mod(i, 2 * n)
            return np.where(m < n, m, 2 * n - 1 - m)
        if self.address == "WRAP":
            return np.mod(i, n)
        return np.clip(i, 0, n - 1)


def sample_trilinear(tex, smp, u, v, w, off):
    X, Y, Z = tex.dims
    tx, ty, tz = u * F32(X) - F32(0.5), v * F32(Y) - F32(0.5), w * F32(Z) - F32(0.5)
    flx, fly, flz = np.floor(tx), np.floor(ty), np.floor(tz)
    fx, fy, fz = (tx - flx).astype(F32), (ty - fly).astype(F32), (tz - flz).astype(F32)
    if ALT["filter_bits"]:                      # fixed-point blend weights (round to nearest)
        q = F32(2.0 ** ALT["filter_bits"])
        fx, fy, fz = (np.floor(fx * q + F32(0.5)) / q).astype(F32), (np.floor(fy * q + F32(0.5)) / q).astype(F32), (np.floor(fz * q + F32(0.5)) / q).astype(F32)
    with np.errstate(invalid="ignore"):
        ix = np.nan_to_num(flx).astype(np.int64) + off[0]
        iy = np.nan_to_num(fly).astype(np.int64) + off[1]
        iz = np.nan_to_num(flz).astype(np.int64) + off[2]
    x0, x1 = smp.tap(ix, X), smp.tap(ix + 1, X)
    y0, y1 = smp.tap(iy, Y), smp.tap(iy + 1, Y)
    z0, z1 = smp.tap(iz, Z), smp.tap(iz + 1, Z)
    d = tex.data

    def lerp(a, b, f):
        return fma32(np.broadcast_to(f[:, None], a.shape).astype(F32), (b - a).astype(F32), a)
    c00 = lerp(d[z0, y0, x0], d[z0, y0, x1], fx)
    c10 = lerp(d[z0, y1, x0], d[z0, y1, x1], fx)
    c01 = lerp(d[z1, y0, x0], d[z1, y0, x1], fx)
    c11 = lerp(d[z1, y1, x0], d[z1, y1, x1], fx)
    r = lerp(lerp(c00, c10, fy), lerp(c01, c11, fy), fz)
    C = r.shape[-1]
    out = np.zeros((len(u), 4), F32)
    out[:, :C] = r
    if C == 3:
        out[:, 3] = 1.0
    return out


class CubePoint:
    """TextureCube sampled at texel-centre directions: returns the texel the direction hits (the reference's
    CSSHCubeMap fetches exactly at texel centres, where bilinear weights are 1/0)."""

    def __init__(self, cube):          # [6][N][N][3]
        self.cube = np.ascontiguousarray(cube, F32)

    def sample(self, d):
        ax, ay, az = np.abs(d[:, 0]), np.abs(d[:, 1]), np.abs(d[:, 2])
        N = self.cube.shape[1]
        face = np.zeros(len(d), np.int64)
        sc = np.zeros(len(d), np.float64); tc = np.zeros(len(d), np.float64); ma = np.ones(len(d), np.float64)
        x, y, z = (d[:, i].astype(np.float64) for i in range(3))
        mx = (ax >= ay) & (ax >= az)
        my = ~mx & (ay >= az)
        mz = ~mx & ~my
        # D3D cube face selection: +X: (-z, -y), -X: (z, -y), +Y: (x, z), -Y: (x, -z), +Z: (x, -y), -Z: (-x, -y)
        for m, pos, fpos, fneg, s_p, t_p, s_n, t_n, mag in (
                (mx, x, 0, 1, -z, -y, z, -y, ax), (my, y, 2, 3, x, z, x, -z, ay), (mz, z, 4, 5, x, -y, -x, -y, az)):
            p = m & (pos >= 0); n_ = m & (pos < 0)
            face[p] = fpos; face[n_] = fneg
            sc[p] = s_p[p]; tc[p] = t_p[p]; sc[n_] = s_n[n_]; tc[n_] = t_n[n_]
            ma[m] = mag[m]
        u = 0.5 * (sc / ma + 1.0); v = 0.5 * (tc / ma + 1.0)
        ix = np.clip(np.floor(u * N).astype(np.int64), 0, N - 1)
        iy = np.clip(np.floor(v * N).astype(np.int64), 0, N - 1)
        out = np.zeros((len(d), 4), F32)
        out[:, :3] = self.cube[face, iy, ix]
        out[:, 3] = 1.0
        return out


class CubeSeamless:
    """TextureCube of RGBA8_UNORM texels [6][N][N][4] with D3D11 addressing, used by the pixel shaders that resolve the
    cube map (gather4 / sample_l): major axis with ties Z > Y > X, per-face (sc, tc) table, bilinear footprint at
    floor(u*N - 0.5), seamless edges.  A footprint texel that falls off one edge is found by FOLDING its centre over the
    cube edge onto the adjacent face (geometric, no adjacency table); a texel off a corner is the mean of the other
    three."""

    def __init__(self, cube):
        """cube: uint8 [6][N][N][4] (RGBA8_UNORM) or float [6][N][N][3|4] (a float format such as the decoded BC6H radiance)"""
        cube = np.asarray(cube)
        if cube.dtype == np.uint8:
            self.t = np.ascontiguousarray(cube, np.uint8)
            self.scale = F32(255.0)
        else:
            c = np.ones(cube.shape[:3] + (4,), F32)
            c[..., :cube.shape[3]] = cube.astype(F32)
            self.t = c
            self.scale = None
        self.N = self.t.shape[1]

    def _fetch(self, f, j, i):
        v = self.t[f, j, i].astype(F32)
        return v / self.scale if self.scale is not None else v

    @property
    def dims(self):
        return (self.N, self.N, 6)

    @staticmethod
    def _select(d):
        x, y, z = d[:, 0], d[:, 1], d[:, 2]
        ax, ay, az = np.abs(x), np.abs(y), np.abs(z)
        mz = (az >= ax) & (az >= ay)
        my = ~mz & (ay >= ax)
        mx = ~mz & ~my
        face = np.zeros(len(d), np.int64)
        sc = np.zeros(len(d), F32); tc = np.zeros(len(d), F32); ma = np.ones(len(d), F32)
        for m, comp, fpos, s_p, t_p, s_n, t_n, mag in ((mx, x, 0, -z, -y, z, -y, ax), (my, y, 2, x, z, x, -z, ay), (mz, z, 4, x, -y, -x, -y, az)):
            p = m & ~(comp < 0); n_ = m & (comp < 0)
            face[p] = fpos; face[n_] = fpos + 1
            sc[p] = s_p[p]; tc[p] = t_p[p]; sc[n_] = s_n[n_]; tc[n_] = t_n[n_]
            ma[m] = mag[m]
        return face, sc, tc, ma

    @staticmethod
    def _point(face, sc, tc):
        one = np.ones_like(sc)
        P = np.zeros((len(sc), 3), F32)
        for f, (a, b, c) in enumerate(((one, -tc, -sc), (-one, -tc, sc), (sc, one, tc), (sc, -one, -tc), (sc, -tc, one), (-sc, -tc, -one))):
            k = face == f
            P[k, 0] = a[k]; P[k, 1] = b[k]; P[k, 2] = c[k]
        return P

    def _texels(self, face, i, j):
        """[n][4] float texels at integer (i, j) of `face`; i/j may be -1 or N (one of them: folded; both: NaN marker)"""
        N = self.N
        out = np.zeros((len(face), 4), F32)
        oi = (i < 0) | (i >= N); oj = (j < 0) | (j >= N)
        inside = ~oi & ~oj
        out[inside] = self._fetch(face[inside], j[inside], i[inside])
        edge = oi ^ oj
        if edge.any():
            f = face[edge]
            sc = ((2 * i[edge] + 1).astype(F32) / F32(N) - F32(1.0)).astype(F32)      # texel centre on the extended face plane
            tc = ((2 * j[edge] + 1).astype(F32) / F32(N) - F32(1.0)).astype(F32)
            P = self._point(f, sc, tc)
            ax = f >> 1
            rows = np.arange(len(f))
            P = np.nan_to_num(P, nan=0.0, posinf=4.0, neginf=-4.0)  # lanes whose pixel was discarded carry garbage
            oa = np.argmax(np.abs(P), axis=1)                     # the one coordinate beyond the cube
            excess = np.abs(P[rows, oa]) - 1
            sign_f = np.sign(P[rows, ax])
            P[rows, oa] = np.sign(P[rows, oa])                    # back onto the cube ...
            P[rows, ax] = sign_f * (1 - excess)                   # ... and down the adjacent face by the same distance
            g, s2, t2, _ = self._select(P.astype(F32))
            i2 = np.clip(np.floor((s2 * F32(0.5) + F32(0.5)) * N).astype(np.int64), 0, N - 1)
            j2 = np.clip(np.floor((t2 * F32(0.5) + F32(0.5)) * N).astype(np.int64), 0, N - 1)
            out[edge] = self._fetch(g, j2, i2)
        corner = oi & oj
        out[corner] = np.nan
        return out, corner

    def footprint(self, d):
        N = self.N
        face, sc, tc, ma = self._select(d.astype(F32))
        u = (F32(0.5) * (sc / ma) + F32(0.5)).astype(F32)
        v = (F32(0.5) * (tc / ma) + F32(0.5)).astype(F32)
        tu = fma32(u, F32(N), F32(-0.5)); tv = fma32(v, F32(N), F32(-0.5))
        i0 = np.floor(tu).astype(np.int64); j0 = np.floor(tv).astype(np.int64)
        fu = (tu - np.floor(tu)).astype(F32); fv = (tv - np.floor(tv)).astype(F32)
        taps = []
        corners = []
        for di, dj in ((0, 1), (1, 1), (1, 0), (0, 0)):          # gather order x, y, z, w
            t, c = self._texels(face, i0 + di, j0 + dj)
            taps.append(t); corners.append(c)
        taps = np.stack(taps, 1)                                  # [n][4 taps][4 channels]
        for k in range(4):
            c = corners[k]
            if c.any():
                others = [q for q in range(4) if q != k]
                acc = taps[c][:, others[0]] + taps[c][:, others[1]]
                acc = (acc + taps[c][:, others[2]]).astype(F32)
                taps[c, k] = (acc / F32(3.0)).astype(F32)
        return taps, fu, fv

    def gather(self, d, channel):
        taps, _, _ = self.footprint(d)
        return np.ascontiguousarray(taps[:, :, channel])

    def sample(self, d):
        taps, fu, fv = self.footprint(d)

        def lerp(a, b, f):
            return fma32(f[:, None], (b - a).astype(F32), a)
        return lerp(lerp(taps[:, 3], taps[:, 2], fu), lerp(taps[:, 0], taps[:, 1], fu), fv)


class Machine:
    def __init__(self, blob, groups, resources, cbs, samplers=None):
        """groups = (gx, gy, gz) thread groups; resources: {'t0': Texture|Structured|CubePoint, 'u0': ...};
        cbs: {slot: ndarray[n][4] uint32}; samplers: {'s0': Sampler}"""
        self.version, self.ins = dxbc.decode(blob)
        self.res = resources
        self.cbs = {k: np.ascontiguousarray(v, U32) for k, v in cbs.items()}
        self.smp = samplers or {}
        tg = [i for i in self.ins if i.op == "DCL_THREAD_GROUP"][0].extra
        self.tg = tuple(int(v) for v in tg)
        gx, gy, gz = groups
        tx, ty, tz = self.tg
        G = np.stack(np.meshgrid(np.arange(gz), np.arange(gy), np.arange(gx), indexing="ij"), -1).reshape(-1, 3)[:, ::-1]
        T = np.stack(np.meshgrid(np.arange(tz), np.arange(ty), np.arange(tx), indexing="ij"), -1).reshape(-1, 3)[:, ::-1]
        gid = np.repeat(G, len(T), axis=0)
        tid = np.tile(T, (len(G), 1))
        self.N = len(gid)
        z4 = np.zeros((self.N, 1), np.int64)
        self.v = {
            "vThreadGroupID": np.concatenate([gid, z4], 1).astype(U32),
            "vThreadIDInGroup": np.concatenate([tid, z4], 1).astype(U32),
            "vThreadID": np.concatenate([gid * np.array(self.tg) + tid, z4], 1).astype(U32),
            "vThreadIDInGroupFlattened": np.repeat((tid[:, 2] * ty * tx + tid[:, 1] * tx + tid[:, 0])[:, None], 4, 1).astype(U32),
        }
        self.group_index = np.repeat(np.arange(len(G)), len(T))
        ntemps = [i for i in self.ins if i.op == "DCL_TEMPS"]
        self.r = np.zeros(((ntemps[0].extra[0] if ntemps else 0) + 1, self.N, 4), U32)
        self.x = {}
        for i in self.ins:
            if i.op == "DCL_INDEXABLE_TEMP":
                idx, size, _ = i.extra
                self.x[int(idx)] = np.zeros((self.N, int(size), 4), U32)
        self.gsm = {}
        for i in self.ins:
            if i.op == "DCL_THREAD_GROUP_SHARED_MEMORY_STRUCTURED":
                stride, count = i.extra
                self.gsm[i.operands[0].indices[0]] = np.zeros((len(G), int(count), int(stride) // 4), U32)
        self.executed = 0

    # ---- operand access -------------------------------------------------------------------------
    def _index(self, ix):
        if isinstance(ix, tuple):
            rel = self.read(ix[1])[:, 0].astype(np.int64)
            return rel + ix[0]
        return ix

    def read(self, o, integer=False):
        t = o.type
        if t == "l":
            imm = o.imm if len(o.imm) == 4 else (o.imm[0],) * 4
            val = np.broadcast_to(np.array(imm, U32), (self.N, 4))
        elif t == "r":
            val = self.r[o.indices[0]]
        elif t == "x":
            arr = self.x[o.indices[0]]
            i = self._index(o.indices[1])
            val = arr[np.arange(self.N), i] if isinstance(i, np.ndarray) else arr[:, i]
        elif t == "cb":
            cb = self.cbs[o.indices[0]]
            i = self._index(o.indices[1])
            val = cb[i] if isinstance(i, np.ndarray) else np.broadcast_to(cb[i], (self.N, 4))
        elif t == "icb":                                  # immediate constant buffer (the CUSTOMDATA block of class 3)
            i = self._index(o.indices[-1])
            val = self.icb[i] if isinstance(i, np.ndarray) else np.broadcast_to(self.icb[i], (self.N, 4))
        elif t in self.v:
            val = self.v[t]
        elif t == "v":                                    # pixel-shader input register
            val = self.inputs[o.indices[0]]
        elif t == "null":
            return None
        else:
            raise NotImplementedError("operand type " + t)
        if o.ncomp == 4 and o.sel in ("swizzle", "select1"):
            val = val[:, list(o.swizzle)]
        elif o.ncomp == 1:
            val = np.broadcast_to(val[:, :1], (self.N, 4))
        if o.modifier:
            if integer:                       # integer instructions: the neg modifier is two's complement negation
                assert o.modifier == 1
                val = ((~np.ascontiguousarray(val).astype(np.uint64) + 1) & 0xFFFFFFFF).astype(U32)
            else:
                f = u2f(np.ascontiguousarray(val))
                if o.modifier & 2:
                    f = np.abs(f)
                if o.modifier & 1:
                    f = -f
                val = f2u(f)
        return np.ascontiguousarray(val)

    def write(self, o, val, mask, sat=False):
        if o.type == "null":
            return
        if sat:
            f = u2f(val)
            with np.errstate(invalid="ignore"):
                f = np.where(f > 0, np.minimum(f, F32(1.0)), F32(0.0)).astype(F32)
            val = f2u(f)
        comps = [c for c in range(4) if (o.mask >> c) & 1] if o.ncomp == 4 else [0]
        if o.type == "r":
            dst = self.r[o.indices[0]]
            for c in comps:
                dst[mask, c] = val[mask, c]
        elif o.type == "o":                                  # pixel-shader output register
            dst = self.outputs.setdefault(o.indices[0], np.zeros((self.N, 4), U32))
            for c in comps:
                dst[mask, c] = val[mask, c]
        elif o.type == "x":
            arr = self.x[o.indices[0]]
            i = self._index(o.indices[1])
            rows = np.arange(self.N)[mask]
            ii = i[mask] if isinstance(i, np.ndarray) else np.full(len(rows), i)
            for c in comps:
                arr[rows, ii, c] = val[mask, c]
        else:
            raise NotImplementedError("dest type " + o.type)

    # ---- run ----------------------------------------------------------------------------------------
    def run(self, max_instructions=5_000_000):
        ins = self.ins
        # pre-match structured control flow
        match = {}
        stack = []
        for pc, i in enumerate(ins):
            if i.op in ("IF", "LOOP", "SWITCH"):
                stack.append([pc])
            elif i.op in ("ELSE", "CASE", "DEFAULT"):
                stack[-1].append(pc)
            elif i.op in ("ENDIF", "ENDLOOP", "ENDSWITCH"):
                grp = stack.pop()
                grp.append(pc)
                for p in grp:
                    match[p] = grp
        N = self.N
        alive = np.ones(N, bool)
        frames = [{"kind": "top", "mask": alive.copy()}]
        pc = 0

        def cur():
            return frames[-1]["mask"]

        def remove(threads, down_to_kind):
            """drop `threads` from every frame from the top down to (and including) the innermost loop/switch"""
            for f in reversed(frames):
                f["mask"] = f["mask"] & ~threads
                if f["kind"] in down_to_kind:
                    f["broke"] = f["broke"] | threads
                    return
        while pc < len(ins):
            i = ins[pc]
            op = i.op
            m = cur()
            self.executed += 1
            if self.executed > max_instructions:
                raise RuntimeError("instruction budget exceeded")
            if op.startswith("DCL") or op in ("CUSTOMDATA", "NOP", "SYNC"):
                pc += 1
                continue
            if op == "IF":
                c = self.read(i.operands[0])[:, 0] != 0
                if not i.test_nz:
                    c = ~c
                frames.append({"kind": "if", "mask": m & c, "entry": m.copy(), "cond": c})
                pc += 1
                continue
            if op == "ELSE":
                f = frames[-1]
                # threads that entered the IF, failed the condition and are still alive in the parent
                f["mask"] = f["entry"] & ~f["cond"] & frames[-2]["mask"]
                pc += 1
                continue
            if op == "ENDIF":
                frames.pop()
                pc += 1
                continue
            if op == "LOOP":
                frames.append({"kind": "loop", "mask": m.copy(), "broke": np.zeros(N, bool), "start": pc})
                pc += 1
                continue
            if op == "ENDLOOP":
                f = frames[-1]
                if f["mask"].any():
                    pc = f["start"] + 1
                else:
                    frames.pop()
                    pc += 1
                continue
            if op in ("BREAK", "BREAKC"):
                t = m
                if op == "BREAKC":
                    c = self.read(i.operands[0])[:, 0] != 0
                    if not i.test_nz:
                        c = ~c
                    t = m & c
                remove(t, ("loop", "switch"))
                pc += 1
                continue
            if op == "SWITCH":
                sel = self.read(i.operands[0])[:, 0].copy()
                frames.append({"kind": "switch", "mask": np.zeros(N, bool), "entry": m.copy(), "sel": sel,
                               "taken": np.zeros(N, bool), "broke": np.zeros(N, bool)})
                pc += 1
                continue
            if op == "CASE":
                f = frames[-1]
                val = i.operands[0].imm[0]
                hit = f["entry"] & (f["sel"] == U32(val)) & ~f["taken"]
                f["taken"] |= hit
                f["mask"] = (f["mask"] | hit) & ~f["broke"]
                pc += 1
                continue
            if op == "DEFAULT":
                f = frames[-1]
                hit = f["entry"] & ~f["taken"]
                f["taken"] |= hit
                f["mask"] = (f["mask"] | hit) & ~f["broke"]
                pc += 1
                continue
            if op == "ENDSWITCH":
                frames.pop()
                pc += 1
                continue
            if op in ("RET", "RETC", "DISCARD"):
                t = m
                if op == "DISCARD":
                    c = self.read(i.operands[0])[:, 0] != 0
                    if not i.test_nz:
                        c = ~c
                    t = m & c
                    self.discarded = getattr(self, "discarded", np.zeros(N, bool)) | t
                if op == "RETC":
                    c = self.read(i.operands[0])[:, 0] != 0
                    if not i.test_nz:
                        c = ~c
                    t = m & c
                for f in frames:
                    f["mask"] = f["mask"] & ~t
                    if "entry" in f:
                        f["entry"] = f["entry"] & ~t
                if not frames[0]["mask"].any() and len(frames) == 1:
                    break
                pc += 1
                continue
            if m.any():
                self.exec_alu(i, m)
            pc += 1
        return self

    # ---- ALU / memory instructions ------------------------------------------------------------------
    def exec_alu(self, i, m):
        op = i.op
        O = i.operands
        INT_OPS = ("IADD", "IMAD", "IMUL", "ISHL", "UDIV", "UMAX", "UMIN", "ULT", "UGE", "AND", "OR", "UTOF", "ITOF", "INEG")

        def R(o):
            return self.read(o, integer=op in INT_OPS)

        def F(k):
            return u2f(R(O[k]))

        def wf(val):
            self.write(O[0], f2u(np.ascontiguousarray(val, F32)), m, i.sat)

        def wu(val):
            self.write(O[0], np.ascontiguousarray(val).astype(U32), m)

        def boolmask(b):
            return np.where(b, U32(0xFFFFFFFF), U32(0))
        with np.errstate(all="ignore"):
            if op == "MOV":
                self.write(O[0], R(O[1]), m, i.sat)
            elif op == "MOVC":
                wu(np.where(R(O[1]) != 0, R(O[2]), R(O[3])))
            elif op == "ADD":
                wf(F(1) + F(2))
            elif op == "MUL":
                wf(F(1) * F(2))
            elif op == "DIV":
                wf(F(1) / F(2))
            elif op == "MAD":
                wf(mad32(F(1), F(2), F(3)))
            elif op in ("DP2", "DP3", "DP4"):
                n = int(op[2])
                a, b = F(1), F(2)
                acc = a[:, 0] * b[:, 0]
                for k in range(1, n):
                    acc = mad32(a[:, k], b[:, k], acc)
                wf(np.repeat(acc[:, None], 4, 1))
            elif op == "MAX":
                wf(np.fmax(F(1), F(2)))
            elif op == "MIN":
                wf(np.fmin(F(1), F(2)))
            elif op == "SQRT":
                wf(np.sqrt(F(1)))
            elif op == "RSQ":
                wf(_ulps((1.0 / np.sqrt(F(1).astype(np.float64))).astype(F32), ALT["rsq_ulps"]))
            elif op == "EXP":
                wf(_ulps(np.exp2(F(1).astype(np.float64)).astype(F32), ALT["exp_ulps"]))
            elif op == "LT":
                wu(boolmask(F(1) < F(2)))
            elif op == "GE":
                wu(boolmask(F(1) >= F(2)))
            elif op == "EQ":
                wu(boolmask(F(1) == F(2)))
            elif op == "NE":
                wu(boolmask(F(1) != F(2)))
            elif op == "AND":
                wu(R(O[1]) & R(O[2]))
            elif op == "OR":
                wu(R(O[1]) | R(O[2]))
            elif op == "IADD":
                wu((R(O[1]).astype(np.int64) + R(O[2]).view(np.int32).astype(np.int64)) & 0xFFFFFFFF)
            elif op == "ISHL":
                wu((R(O[1]).astype(np.uint64) << (R(O[2]) & 31).astype(np.uint64)) & 0xFFFFFFFF)
            elif op == "USHR":
                wu((R(O[1]).astype(np.uint64) >> (R(O[2]) & 31).astype(np.uint64)) & 0xFFFFFFFF)
            elif op == "IMAD":
                a, b, c = (R(O[k]).view(np.int32).astype(np.int64) for k in (1, 2, 3))
                wu((a * b + c) & 0xFFFFFFFF)
            elif op == "IMUL":
                a, b = (R(O[k]).view(np.int32).astype(np.int64) for k in (2, 3))
                p = a * b
                self.write(O[0], ((p >> 32) & 0xFFFFFFFF).astype(U32), m)
                self.write(O[1], (p & 0xFFFFFFFF).astype(U32), m)
            elif op == "UDIV":
                a, b = R(O[2]).astype(np.uint64), R(O[3]).astype(np.uint64)
                q = np.where(b == 0, 0xFFFFFFFF, a // np.maximum(b, 1))
                r = np.where(b == 0, 0xFFFFFFFF, a % np.maximum(b, 1))
                self.write(O[0], q.astype(U32), m)
                self.write(O[1], r.astype(U32), m)
            elif op == "UMAX":
                wu(np.maximum(R(O[1]), R(O[2])))
            elif op == "UMIN":
                wu(np.minimum(R(O[1]), R(O[2])))
            elif op == "ULT":
                wu(boolmask(R(O[1]) < R(O[2])))
            elif op == "UGE":
                wu(boolmask(R(O[1]) >= R(O[2])))
            elif op == "UTOF":
                wf(R(O[1]).astype(F32))
            elif op == "ITOF":
                wf(R(O[1]).view(np.int32).astype(F32))
            elif op == "NOT":
                wu(~R(O[1]))
            elif op == "INEG":
                wu((~R(O[1]).astype(np.uint64) + 1) & 0xFFFFFFFF)
            elif op == "FRC":
                a = F(1)
                wf(a - np.floor(a))
            elif op == "GATHER4":
                res = self.res["t" + str(O[2].indices[0])]
                chan = O[3].swizzle[0]                            # sampler operand carries the channel select
                g = res.gather(F(1)[:, :3], chan)
                self.write(O[0], f2u(g)[:, list(O[2].swizzle)], m)
            elif op == "RESINFO":
                res = self.res[O[2].type + str(O[2].indices[0])]
                X, Y, Z = res.dims
                rt = i.ctrl & 3                                   # 0 float, 1 rcp float, 2 uint
                dims = np.array([X, Y, Z, 1])
                if rt == 2:
                    val = np.broadcast_to(dims.astype(U32), (self.N, 4))
                else:
                    fd = dims.astype(F32) if rt == 0 else (F32(1.0) / dims.astype(F32))
                    val = np.broadcast_to(f2u(fd), (self.N, 4))
                val = np.ascontiguousarray(val)[:, list(O[2].swizzle)]
                self.write(O[0], val, m)
            elif op in ("LD", "LD_UAV_TYPED"):
                res = self.res[O[2].type + str(O[2].indices[0])]
                a = R(O[1]).view(np.int32).astype(np.int64)
                t = res.load(a[:, 0], a[:, 1], a[:, 2], m)
                self.write(O[0], f2u(t)[:, list(O[2].swizzle)], m)
            elif op == "STORE_UAV_TYPED":
                res = self.res["u" + str(O[0].indices[0])]
                a = R(O[1]).view(np.int32).astype(np.int64)
                res.store(a[:, 0], a[:, 1], a[:, 2], u2f(R(O[2])), m)
            elif op == "SAMPLE_L":
                res = self.res["t" + str(O[2].indices[0])]
                c = F(1)
                if isinstance(res, (CubePoint, CubeSeamless)):
                    t = res.sample(c[:, :3]) if isinstance(res, CubeSeamless) else res.sample(c)
                else:
                    t = sample_trilinear(res, self.smp["s" + str(O[3].indices[0])], c[:, 0], c[:, 1], c[:, 2], i.offsets)
                self.write(O[0], f2u(t)[:, list(O[2].swizzle)], m)
            elif op == "LD_STRUCTURED":
                idx = R(O[1])[:, 0].astype(np.int64)
                off = R(O[2])[:, 0].astype(np.int64) // 4
                src = O[3]
                if src.type == "g":
                    words = self.gsm[src.indices[0]]
                    n, stride = words.shape[1], words.shape[2]
                    out = np.zeros((self.N, 4), U32)
                    ok = (idx >= 0) & (idx < n)
                    for k in range(4):
                        w = off + k
                        good = ok & (w < stride)
                        out[good, k] = words[self.group_index[good], idx[good], w[good]]
                else:
                    words = self.res[src.type + str(src.indices[0])].words
                    n, stride = words.shape
                    out = np.zeros((self.N, 4), U32)
                    ok = (idx >= 0) & (idx < n)
                    for k in range(4):
                        w = off + k
                        good = ok & (w < stride)
                        out[good, k] = words[idx[good], w[good]]
                self.write(O[0], out[:, list(src.swizzle)], m)
            elif op == "STORE_STRUCTURED":
                dst = O[0]
                idx = R(O[1])[:, 0].astype(np.int64)
                off = R(O[2])[:, 0].astype(np.int64) // 4
                val = R(O[3])
                comps = [c for c in range(4) if (dst.mask >> c) & 1]
                if dst.type == "g":
                    words = self.gsm[dst.indices[0]]
                    ok = m & (idx >= 0) & (idx < words.shape[1])
                    for k, c in enumerate(comps):
                        words[self.group_index[ok], idx[ok], off[ok] + k] = val[ok, c]
                else:
                    words = self.res["u" + str(dst.indices[0])].words
                    ok = m & (idx >= 0) & (idx < words.shape[0])
                    for k, c in enumerate(comps):
                        words[idx[ok], off[ok] + k] = val[ok, c]
            else:
                raise NotImplementedError(op)


class PixelMachine(Machine):
    """ps_5_0: one lane per pixel; `inputs` = {register index: float32[N][4]} (the interpolants at the pixel centres),
    outputs in self.outputs[index] (uint32 bit patterns), self.discarded marks lanes that executed a discard."""

    def __init__(self, blob, inputs, resources, cbs, samplers=None):
        self.version, self.ins = dxbc.decode(blob)
        self.res = resources
        self.cbs = {k: np.ascontiguousarray(v, U32) for k, v in cbs.items()}
        self.smp = samplers or {}
        self.inputs = {k: f2u(np.ascontiguousarray(v, F32)) for k, v in inputs.items()}
        self.N = len(next(iter(inputs.values())))
        self.v = {}
        self.outputs = {}
        self.discarded = np.zeros(self.N, bool)
        ntemps = [i for i in self.ins if i.op == "DCL_TEMPS"]
        self.r = np.zeros(((ntemps[0].extra[0] if ntemps else 0) + 1, self.N, 4), U32)
        self.x = {}
        for i in self.ins:
            if i.op == "DCL_INDEXABLE_TEMP":
                idx, size, _ = i.extra
                self.x[int(idx)] = np.zeros((self.N, int(size), 4), U32)
        self.gsm = {}
        self.executed = 0


class VertexMachine(PixelMachine):
    """vs_5_0: one lane per vertex; `inputs` = {register index: uint32[N][4]} (SV_VertexID, SV_InstanceID ... as the input assembler
    hands them over: raw bits), outputs in self.outputs[index].  Immediate constant buffers (icb) come from the shader's CUSTOMDATA."""

    def __init__(self, blob, inputs, resources, cbs, samplers=None):
        super().__init__(blob, {k: np.zeros((len(v), 4), F32) for k, v in inputs.items()}, resources, cbs, samplers)
        self.inputs = {k: np.ascontiguousarray(v, U32) for k, v in inputs.items()}
        for i in self.ins:
            if i.op == "CUSTOMDATA" and (i.ctrl & 0x1FFFFF) == 3:
                self.icb = np.array(i.raw, U32).reshape(-1, 4)


def run_vertex_shader(path, inputs, cbs, **kw):
    m = VertexMachine(open(path, "rb").read(), inputs, {}, cbs)
    return m.run(**kw)


def run_pixel_shader(path, inputs, resources, cbs, samplers=None, **kw):
    m = PixelMachine(open(path, "rb").read(), inputs, resources, cbs, samplers)
    return m.run(**kw)


def run_shader(path, groups, resources, cbs, samplers=None, **kw):
    blob = open(path, "rb").read()
    return Machine(blob, groups, resources, cbs, samplers).run(**kw)
