"""k_bc6h_decode + the DDS parser behind fx_dds_decode_cube (row f-4) against the oracle, bit for bit."""
import os
import struct

import numpy as np
import pytest

import fluidx12_amd as fx
from oracle import orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIX = np.load(os.path.join(GOLD, "bc6h_fixture.npz"))


def make_dds(blocks_per_face, n):
    """a DDS cube map (DX10 header, BC6H_UF16, one mip) around 6 x (n/4)^2 given blocks; header taken from the fixture"""
    hdr = bytearray(FIX["dds_mip3"].tobytes()[:148])
    struct.pack_into("<I", hdr, 12, n)
    struct.pack_into("<I", hdr, 16, n)
    struct.pack_into("<I", hdr, 20, ((n + 3) // 4) ** 2 * 16)
    return bytes(hdr) + np.ascontiguousarray(blocks_per_face, np.uint8).tobytes()


def probe():
    f = fx.Fluid()
    assert f.Init(64, 64, (16, 16, 16))
    return fx.LightProbe(f)


@pytest.mark.parametrize("n,seed", [(64, 1), (4, 2), (20, 3)])
def test_random_blocks_equal_oracle(n, seed):
    """every 128-bit pattern is a block: random bytes reach all 14 modes, all 32 partitions and the reserved modes"""
    nb = (n + 3) // 4
    blocks = np.random.default_rng(seed).integers(0, 256, (6, nb * nb, 16), dtype=np.uint8)
    dds = make_dds(blocks, n)
    want, hist = orc.dds_bc6h_cube(dds, 0)
    if n >= 64:
        assert (hist > 0).all()
    got = probe().decode_dds(dds)
    assert got.shape == (6, n, n, 3)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_fixture_asset_mip_and_its_sh():
    """real encoder output (mip 3 of the reference's rnl_cross.dds): decode == oracle == committed cube, and
    LightProbe.Init(<dds bytes>) + TransformSH reproduces the oracle's SH of it"""
    dds = FIX["dds_mip3"].tobytes()
    p = probe()
    got = p.decode_dds(dds)
    assert np.array_equal(got, FIX["cube_mip3"])
    assert p.Init(dds)
    p.TransformSH()
    sh = p.GetSH()
    assert np.allclose(sh, FIX["sh_mip3"], rtol=2e-5, atol=2e-6)
    assert sh[0].min() > 0.5                                      # the probe is bright: L00 of an HDR sky


def test_malformed_containers_are_rejected():
    p = probe()
    dds = bytearray(FIX["dds_mip3"].tobytes())
    assert p.decode_dds(bytes(dds[:147])) is None                 # truncated header
    assert p.decode_dds(bytes(dds[:-16])) is None                 # truncated payload
    bad = bytearray(dds); bad[128] = 98                           # DXGI_FORMAT_BC7_UNORM
    assert p.decode_dds(bytes(bad)) is None
    bad = bytearray(dds); bad[0] = ord("X")
    assert p.decode_dds(bytes(bad)) is None
    assert p.decode_dds(bytes(dds), mip=1) is None                # no such mip
    assert p.Init(bytes(bad)) is False


def dds_cube(cube, fmt, mips=1, legacy=False):
    """a DDS cube map around cube[6][n][n][3] floats in an uncompressed format (with its mip chain made by 2 x 2 box filters)"""
    n = cube.shape[1]
    chain = [cube]
    for _ in range(1, mips):
        c = chain[-1]
        chain.append(c.reshape(6, c.shape[1] // 2, 2, c.shape[2] // 2, 2, 3).mean(axis=(2, 4)).astype(np.float32))

    def enc(c):
        rgba = np.concatenate([c, np.ones(c.shape[:-1] + (1,), np.float32)], axis=-1)
        if fmt == "rgba32f":
            return rgba.astype("<f4").tobytes()
        if fmt == "rgb32f":
            return c.astype("<f4").tobytes()
        if fmt == "rgba16f":
            return rgba.astype("<f2").tobytes()
        return np.rint(np.clip(rgba, 0, 1) * 255).astype(np.uint8).tobytes()

    hdr = bytearray(128)
    hdr[0:4] = b"DDS "
    struct.pack_into("<IIIII", hdr, 4, 124, 0x1007 | (0x20000 if mips > 1 else 0), n, n, 0)
    struct.pack_into("<I", hdr, 28, mips)
    struct.pack_into("<II", hdr, 76, 32, 4)                            # DDS_PIXELFORMAT: size, DDPF_FOURCC
    struct.pack_into("<II", hdr, 108, 0x1008 | (0x400000 if mips > 1 else 0), 0xFE00)   # caps: complex | texture (| mipmap); caps2: cube map, all faces
    if legacy:
        struct.pack_into("<I", hdr, 84, {"rgba16f": 113, "rgba32f": 116}[fmt])
        extra = b""
    else:
        hdr[84:88] = b"DX10"
        extra = struct.pack("<IIIII", {"rgba32f": 2, "rgb32f": 6, "rgba16f": 10, "rgba8": 28}[fmt], 3, 4, 1, 0)
    body = b"".join(b"".join(enc(c[f]) for c in chain) for f in range(6))
    return bytes(hdr) + extra + body, chain


@pytest.mark.parametrize("fmt,legacy", [("rgba32f", False), ("rgb32f", False), ("rgba16f", False), ("rgba8", False), ("rgba16f", True), ("rgba32f", True)])
def test_uncompressed_dds_cubes_decode(fmt, legacy):
    """LightProbe::Init hands XUSG's DDS loader whatever cube it is given (LightProbe.cpp:41-46): besides the reference's BC6H asset the
    container parser takes the uncompressed float / unorm formats, DX10 and legacy headers, any mip of the chain, and LightProbe.Init +
    TransformSH run on the result"""
    rng = np.random.default_rng(9)
    cube = (rng.random((6, 16, 16, 3)) * (4.0 if fmt != "rgba8" else 1.0)).astype(np.float32)
    dds, chain = dds_cube(cube, fmt, mips=3, legacy=legacy)
    p = probe()
    for mip in range(3):
        got = p.decode_dds(dds, mip=mip)
        want = chain[mip]
        if fmt == "rgba16f":
            want = want.astype(np.float16).astype(np.float32)
        if fmt == "rgba8":
            want = (np.rint(np.clip(want, 0, 1) * 255).astype(np.uint8).astype(np.float32) / np.float32(255)).astype(np.float32)
        assert got is not None and got.shape == want.shape and np.array_equal(got, want), (fmt, mip)
    assert p.decode_dds(dds, mip=3) is None
    assert p.decode_dds(dds[:-8]) is None                              # truncated payload
    assert p.Init(dds)
    p.TransformSH()
    want_sh = orc.sh_transform(p.decode_dds(dds))
    assert np.allclose(p.GetSH(), want_sh, rtol=1e-5, atol=1e-6)
