"""Independent numpy restatement of the simulation shaders, written from the HLSL *source*
(/root/reference/FluidX12/Content/Shaders/CSAdvect.hlsl, CSProject3D.hlsl, CSProject2D.hlsl,
CSPoisson.hlsli) rather than from the compiled DXBC the C++ oracle follows.  It exists only to
cross-check the oracle (tests/test_oracle.py): two restatements made from two statements of the
reference must agree to rounding (association order differs, so the comparison is ~1e-6, not bit-exact).
Vectorised fp32; never shipped to the product path.
"""
import numpy as np

f32 = np.float32


def _tap(i, n, mirror):
    if mirror:
        m = np.mod(i, 2 * n)
        return np.where(m < n, m, 2 * n - 1 - m)
    return np.clip(i, 0, n - 1)


def trilinear(field, u, v, w, mirror=False):
    """field [Z][Y][X] (+ trailing channel dims); u,v,w normalised coords (arrays)."""
    Z, Y, X = field.shape[:3]
    out = None
    tx, ty, tz = u * f32(X) - f32(0.5), v * f32(Y) - f32(0.5), w * f32(Z) - f32(0.5)
    ix, iy, iz = np.floor(tx), np.floor(ty), np.floor(tz)
    fx, fy, fz = (tx - ix).astype(f32), (ty - iy).astype(f32), (tz - iz).astype(f32)
    ix, iy, iz = ix.astype(np.int64), iy.astype(np.int64), iz.astype(np.int64)
    extra = field.ndim - 3
    ex = (slice(None),) * 0
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                wgt = (fx if dx else 1 - fx) * (fy if dy else 1 - fy) * (fz if dz else 1 - fz)
                val = field[_tap(iz + dz, Z, mirror), _tap(iy + dy, Y, mirror), _tap(ix + dx, X, mirror)]
                if extra:
                    wgt = wgt[..., None]
                out = val * wgt if out is None else out + val * wgt
    return out.astype(f32)


def advect(vel, col, dt, mirror=False):
    """CSAdvect.hlsl:41-79.  vel (3,Z,Y,X), col (Z,Y,X,4)."""
    _, Z, Y, X = vel.shape
    dt = f32(dt)
    z, y, x = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
    px = ((x + 0.5) / X).astype(f32)
    py = ((y + 0.5) / Y).astype(f32)
    pz = ((z + 0.5) / Z).astype(f32)
    ax, ay, az = px - vel[0] * dt, py - vel[1] * dt, pz - vel[2] * dt
    u = np.stack([trilinear(vel[a], ax, ay, az, mirror) for a in range(3)])
    c = trilinear(col, ax, ay, az, mirror)
    dx, dy, dz = px - f32(0.5), py - f32(0.1), pz - f32(0.5)
    is3d = Z > 1
    r = f32(1.0 / 16.0) if is3d else f32(1.0 / 32.0)
    basis = np.exp(f32(-4.0) * (dx * dx + dy * dy + dz * dz) / (r * r)).astype(f32)
    inside = basis >= f32(np.exp(-4.0))
    ext = np.stack([np.zeros_like(basis), f32(48.0) * basis, np.zeros_like(basis)])
    if is3d:
        ext = ext * f32(4.0) + np.stack([-dz, np.zeros_like(dz), dx]) * f32(200.0)
    u = np.where(inside[None], u + ext * dt, u)
    imp = np.array([0.2, 0.4, 1.0, 1.0], f32) * f32(40.0)
    c = np.where(inside[..., None], np.clip(c + imp * dt * basis[..., None], 0, 1), c)
    atten = max(f32(1.0) - f32(0.2) * dt, f32(0.0))
    return (u * atten).astype(f32), (c * atten).astype(f32)


def _shift(a, axis, d):
    """a sampled at index clamp(i + d) along axis."""
    n = a.shape[axis]
    idx = np.clip(np.arange(n) + d, 0, n - 1)
    return np.take(a, idx, axis=axis)


def divergence(vel):
    """GetDivergence, CSProject3D.hlsl:39-50 (2D: CSProject2D.hlsl:37-46)."""
    ux, uy, uz = vel
    d = (_shift(ux, 2, 1) - _shift(ux, 2, -1)) + (_shift(uy, 1, 1) - _shift(uy, 1, -1))
    if vel.shape[1] > 1:
        d = d + (_shift(uz, 0, 1) - _shift(uz, 0, -1))
    return (f32(0.5) * d).astype(f32)


def jacobi(p, b, iters):
    """Synchronous schedule of Poisson(), CSPoisson.hlsli:8-26, fixed sweep count."""
    p = p.astype(f32)
    is3d = p.shape[0] > 1
    n = f32(6.0 if is3d else 4.0)
    for _ in range(iters):
        s = -b + _shift(p, 2, -1) + _shift(p, 2, 1) + _shift(p, 1, -1) + _shift(p, 1, 1)
        if is3d:
            s = s + _shift(p, 0, -1) + _shift(p, 0, 1)
        p = (s / n).astype(f32)
    return p


def project(vel, p):
    """Project + wall factor, CSProject3D.hlsl:55-63,105-112 (2D: CSProject2D.hlsl:51-59,99-105)."""
    _, Z, Y, X = vel.shape
    is3d = Z > 1
    rho = f32(0.48) if is3d else f32(1.0)
    u = vel.copy()
    u[0] -= f32(0.5) * (_shift(p, 2, 1) - _shift(p, 2, -1)) / rho
    u[1] -= f32(0.5) * (_shift(p, 1, 1) - _shift(p, 1, -1)) / rho
    if is3d:
        u[2] -= f32(0.5) * (_shift(p, 0, 1) - _shift(p, 0, -1)) / rho
    z, y, x = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
    pos = [((x + 0.5) / X * 2 - 1).astype(f32), ((y + 0.5) / Y * 2 - 1).astype(f32),
           (((z + 0.5) / Z * 2 - 1) if is3d else np.full(z.shape, 0.5)).astype(f32)]
    for a in range(3):
        fac = np.clip((f32(0.97) - np.abs(pos[a])) / f32(0.03), -1, 1).astype(f32)
        u[a] = np.where(u[a] * pos[a] > 0, u[a] * fac, u[a])
    return u.astype(f32)


def step(vel, col, p, dt, iters, mirror=False):
    v1, c1 = advect(vel, col, dt, mirror)
    b = divergence(v1)
    p = jacobi(p, b, iters)
    return project(v1, p), c1, p
