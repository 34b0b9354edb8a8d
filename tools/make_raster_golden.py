#!/usr/bin/env python3
"""The reference's SHIPPED cube resolve -- Fluid::renderCube (Fluid.cpp:910-931): `Draw(4, 6)` of VSCube.cso + PSCube.cso as triangle
strips with front-face culling and the PREMULTIPLIED blend -- executed on the CPU: the two shader binaries through tools/dxbc_interp.py
and, between them, a restatement of the fixed-function stages D3D11 defines (functional spec 3.4.3 / 3.4.2): window coordinates snapped to
1/256 pixel, the top-left fill rule, clockwise = front facing, perspective-correct interpolation of the interpolants at pixel centres.

Why ("missing" 3): the product resolves the cube map in the raster-free formulation the reference also carries
(PSRayCastCube.hlsl; row f-1) and is pinned bit for bit to PSRayCastCube.cso's outputs -- but what the reference's executable draws is
this path, and whether the two agree at the cube's silhouette had never been looked at.  Output: tests/golden/dxbc_raster.npz (inputs =
those of dxbc_resolve.npz: the same cube maps, frame constants, 160 x 120 target) -- per pixel the coverage of the rasterised interior
faces, PSCube.cso's SV_TARGET and its discards; tests/test_dxbc_golden.py compares it with the raster-free goldens.

What D3D leaves open here and this file chooses: interpolation in fp64 from the snapped vertex positions, rounded once to fp32 (hardware
interpolators carry less; the spec bounds none of it).

    python tools/make_raster_golden.py          (CPU only, needs /root/reference/Bin; a few seconds)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import dxbc_interp as di                    # noqa: E402

F32, U32 = np.float32, np.uint32
BIN = "/root/reference/Bin"
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def vertices(cb0):
    """VSCube.cso for the 6 instances x 4 vertices of Draw(4, 6): clip-space position, UVW, LPt"""
    vid = np.tile(np.arange(4, dtype=U32), 6)
    iid = np.repeat(np.arange(6, dtype=U32), 4)
    z = np.zeros(24, U32)
    m = di.run_vertex_shader(os.path.join(BIN, "VSCube.cso"), {0: np.stack([vid, z, z, z], 1), 1: np.stack([iid, z, z, z], 1)}, {0: cb0.view(U32)})
    pos = m.outputs[0].view(F32).reshape(6, 4, 4)
    uvw = m.outputs[1].view(F32).reshape(6, 4, 4)[..., :3]
    lpt = m.outputs[2].view(F32).reshape(6, 4, 4)[..., :3]
    return pos, uvw, lpt


def rasterise(pos, attrs, W, H):
    """-> covered[H, W], face[H, W], interpolated attrs[H, W, K] of the triangles that survive CULL_FRONT (clockwise = front)"""
    covered = np.zeros((H, W), bool)
    hits = np.zeros((H, W), np.int32)
    face = np.full((H, W), -1, np.int32)
    out = np.zeros((H, W, attrs.shape[-1]), np.float64)
    py, px = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    cx, cy = (px * 256 + 128).astype(np.int64), (py * 256 + 128).astype(np.int64)        # pixel centres in 1/256 pixel
    assert (pos[..., 3] > 0).all(), "a vertex behind the eye: this restatement has no clipper"
    ndc = pos[..., :3].astype(np.float64) / pos[..., 3:4].astype(np.float64)
    sx = np.rint((ndc[..., 0] * 0.5 + 0.5) * W * 256).astype(np.int64)                    # snapped window coordinates
    sy = np.rint((0.5 - ndc[..., 1] * 0.5) * H * 256).astype(np.int64)
    invw = 1.0 / pos[..., 3].astype(np.float64)
    for f in range(6):
        for tri in ((0, 1, 2), (2, 1, 3)):                            # the strip's two triangles, winding kept
            a, b, c = tri
            area2 = (sx[f, b] - sx[f, a]) * (sy[f, c] - sy[f, a]) - (sy[f, b] - sy[f, a]) * (sx[f, c] - sx[f, a])
            if area2 >= 0:                                            # clockwise on a y-down window: front facing -> culled (and degenerate)
                continue
            v = [a, c, b]                                             # re-ordered clockwise: every edge function is positive inside
            inside = np.ones((H, W), bool)
            E = []
            for k in range(3):
                i0, i1 = v[k], v[(k + 1) % 3]
                dx, dy = sx[f, i1] - sx[f, i0], sy[f, i1] - sy[f, i0]
                e = dx * (cy - sy[f, i0]) - dy * (cx - sx[f, i0])
                top_left = (dy == 0 and dx > 0) or dy < 0              # top edge: horizontal, interior below; left edge: going up
                inside &= (e > 0) | ((e == 0) & top_left)
                E.append(e.astype(np.float64))
            if not inside.any():
                continue
            # barycentrics: the edge function opposite a vertex / the area
            lam = {v[2]: E[0], v[0]: E[1], v[1]: E[2]}
            den = sum(lam[i] * invw[f, i] for i in (a, b, c))
            val = sum(lam[i][..., None] * invw[f, i] * attrs[f, i].astype(np.float64) for i in (a, b, c)) / den[..., None]
            out[inside] = val[inside]
            hits[inside] += 1
            covered |= inside
            face[inside] = f
    assert hits.max() <= 1, "two interior faces cover one pixel centre"
    return covered, face, out.astype(F32)


def main():
    res = np.load(os.path.join(GOLD, "dxbc_resolve.npz"))
    cb0, cb1 = res["cb_per_object"], res["cb_per_frame"]
    W, H = int(res["params"][0]), int(res["params"][1])
    pos, uvw, lpt = vertices(cb0)
    covered, face, att = rasterise(pos, np.concatenate([uvw, lpt], -1), W, H)
    out = {"covered": covered, "face": face.astype(np.int8), "params": res["params"], "vs_pos": pos, "vs_uvw": uvw, "vs_lpt": lpt}
    idx = np.nonzero(covered.ravel())[0]
    a = att.reshape(-1, 6)[idx]
    v1 = np.zeros((len(idx), 4), F32); v1[:, :3] = a[:, 0:3]
    v2 = np.zeros((len(idx), 4), F32); v2[:, :3] = a[:, 3:6]
    for name in ("rendered16", "random8"):
        cube = res["cube_" + name]
        m = di.run_pixel_shader(os.path.join(BIN, "PSCube.cso"), {1: v1, 2: v2}, {"t0": di.CubeSeamless(cube)},
                                {0: cb0.view(U32), 1: cb1.view(U32)}, {"s0": di.Sampler("CLAMP")})
        target = np.zeros((H * W, 4), F32)
        disc = np.ones(H * W, bool)
        target[idx] = m.outputs[0].view(F32).reshape(-1, 4)
        disc[idx] = m.discarded
        target[disc] = 0
        out["target_" + name], out["discard_" + name] = target.reshape(H, W, 4), disc.reshape(H, W)
        ref_t, ref_d = res["target_" + name], res["discard_" + name]
        both = ~disc.reshape(H, W) & ~ref_d
        d = np.abs(target.reshape(H, W, 4) - ref_t)[both]
        print("%s: raster covers %d pixels (%d drawn), raster-free draws %d; drawn by one only: %d; on the %d common pixels max |d| %.3e, pixels above 1/255: %d, above 1e-5: %d" % (
            name, covered.sum(), (~disc).sum(), (~ref_d).sum(), (disc.reshape(H, W) != ref_d).sum(), both.sum(), d.max() if d.size else 0,
            (d.max(axis=-1) > 1 / 255).sum() if d.size else 0, (d.max(axis=-1) > 1e-5).sum() if d.size else 0))
    np.savez_compressed(os.path.join(GOLD, "dxbc_raster.npz"), **out)


if __name__ == "__main__":
    main()
