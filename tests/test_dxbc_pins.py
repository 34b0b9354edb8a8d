"""Pins between the oracle and the reference's SHIPPED shader binaries.

tests/golden/cso_constants.json is a digest (tools/make_cso_fixture.py) of /root/reference/Bin/*.cso -- the
DXBC the reference actually runs: opcode histograms, thread-group sizes and the folded fp32 immediates.
The oracle (and the HIP kernels) restate exactly those constants; this test ties the three together.
No GPU.  When /root/reference is mounted (the authoring container) the digest is also regenerated and the
decoder is run over every shipped blob as a self-check."""
import json
import os
import struct
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "cso_constants.json")))


def bits(x):
    return "0x%08x" % struct.unpack("<I", struct.pack("<f", x))[0]


# (shader, fp32 value as the oracle writes it, literal text expected in the oracle source, oracle file)
PINS = [
    ("CSAdvect", 0.0183156393, "0.0183156393f", "orc_sim.cpp"),       # exp(-4) folded            CSAdvect.hlsl:60
    ("CSAdvect", 1.44269502, "1.44269502f", "orc_sim.cpp"),           # log2(e): exp -> exp2       :35
    ("CSAdvect", 0.00390625, "0.00390625f", "orc_sim.cpp"),           # (1/16)^2                   Impulse.hlsli:15
    ("CSAdvect", 0.0009765625, "0.0009765625f", "orc_sim.cpp"),       # (1/32)^2 (2-D)
    ("CSAdvect", 192.0, "192.0f", "orc_sim.cpp"),                     # 48 * 4 folded              CSAdvect.hlsl:11-12
    ("CSAdvect", 48.0, "48.0f", "orc_sim.cpp"),
    ("CSAdvect", 200.0, "200.0f", "orc_sim.cpp"),                     # g_vortScl
    ("CSAdvect", 0.200000003, "0.200000003f", "orc_sim.cpp"),         # g_dissipation
    ("CSAdvect", 8.0, "8.0f", "orc_sim.cpp"),                         # g_impulse = (.2,.4,1,1)*40
    ("CSAdvect", 16.0, "16.0f", "orc_sim.cpp"),
    ("CSAdvect", 40.0, "40.0f", "orc_sim.cpp"),
    ("CSProject3D", None, "0x3e2aaaabu", "orc_sim.cpp"),              # 1/6 as a multiply          CSPoisson.hlsli:19
    ("CSProject3D", None, "0x3f855556u", "orc_sim.cpp"),              # 0.5f / 0.48f               CSProject3D.hlsl:62
    ("CSProject3D", 0.970000029, "0.970000029f", "orc_sim.cpp"),      # wall                       :108
    ("CSProject3D", 33.3333359, "33.3333359f", "orc_sim.cpp"),        # 1 / 0.03 as a multiply
    ("CSProject3D", 0.00100000005, "0.00100000005f", "orc_sim.cpp"),  # early-out threshold        CSPoisson.hlsli:24
    ("CSProject2D", 0.25, "0.25f", "orc_sim.cpp"),
    ("CSProject2D", 33.3333359, "33.3333359f", "orc_sim.cpp"),
    ("CSRayMarchV", 3.46410155, "3.46410155f", "orc_render.cpp"),     # 2 sqrt(3)                  RayMarch.hlsli:29
    ("CSRayMarchV", 0.800000012, "0.800000012f", "orc_render.cpp"),   # ABSORPTION
    ("CSRayMarchV", 0.00999999978, "0.00999999978f", "orc_render.cpp"),   # ZERO_THRESHOLD
    ("CSRayMarchV", 0.159154937, "0.159154937f", "orc_render.cpp"),   # 1 / (2 pi)                 CSRayMarch.hlsl:192
    ("CSRayMarchV", 0.00390625, "0.00390625f", "orc_render.cpp"),     # 1/256                      RayMarch.hlsli:203
    ("CSRayMarchV", 1.5, "1.5f", "orc_render.cpp"),
    ("CSRayMarchV", 3.40282347e+38, "3.40282347e+38f", "orc_render.cpp"),   # FLT_MAX              RayMarch.hlsli:151
    ("CSRayMarchL", 0.429042757, "0.429042757f", "orc_render.cpp"),   # SH irradiance c1..c4       SHIrradianceTypeless.hlsli:18-21
    ("CSRayMarchL", 0.247707963, "0.247707963f", "orc_render.cpp"),
    ("CSRayMarchL", 0.886226952, "0.886226952f", "orc_render.cpp"),
    ("CSRayMarchL", 0.858085513, "0.858085513f", "orc_render.cpp"),
    ("CSRayMarchL", 1.02332675, "1.02332675f", "orc_render.cpp"),
    ("CSRayMarch", 0.429042757, "0.429042757f", "orc_render.cpp"),
    ("CSSHCubeMap", 0.282094806, "0.282094806f", "orc_sh.cpp"),       # sh_eval_basis_2 as compiled SHMath.hlsli:37-66
    ("CSSHCubeMap", 0.488602519, "0.488602519f", "orc_sh.cpp"),
    ("CSSHCubeMap", 1.09254849, "1.09254849f", "orc_sh.cpp"),
    ("CSSHCubeMap", 0.946174681, "0.946174681f", "orc_sh.cpp"),
    ("CSSHCubeMap", 0.546274245, "0.546274245f", "orc_sh.cpp"),
    ("CSSHNormalize", 12.566371, "12.566371f", "orc_sh.cpp"),         # 4 pi                       CSSHNormalize.hlsl:15
]


@pytest.mark.parametrize("shader,value,literal,src", PINS)
def test_oracle_constant_is_the_shipped_one(shader, value, literal, src):
    imms = FIX[shader]["float_immediates"]
    want = literal[:-1] if value is None else bits(value)
    neg = None if value is None else bits(-value)
    assert want in imms or (neg and neg in imms), (shader, want)
    text = open(os.path.join(ROOT, "oracle", src)).read()
    assert literal in text, "oracle/%s no longer uses %s" % (src, literal)


def test_kernels_use_the_same_constants():
    sim = open(os.path.join(ROOT, "fluidx12_amd", "csrc", "fx_sim.hip")).read()
    ren = open(os.path.join(ROOT, "fluidx12_amd", "csrc", "fx_march.h")).read()      # the marches of both render paths
    for path in ("fx_render.hip", "fx_render_accel.hip"):
        assert "0.159154937f" in open(os.path.join(ROOT, "fluidx12_amd", "csrc", path)).read()
    ren += "0.159154937f"
    sh = open(os.path.join(ROOT, "fluidx12_amd", "csrc", "fx_sh.hip")).read()
    for lit in ("0x3e2aaaabu", "0x3f855556u", "0.0183156393f", "1.44269502f", "33.3333359f", "0.970000029f", "192.0f"):
        assert lit in sim, lit
    for lit in ("3.46410155f", "0.800000012f", "0.00999999978f", "0.159154937f", "0.429042757f", "1.02332675f"):
        assert lit in ren, lit
    for lit in ("0.282094806f", "0.946174681f", "12.566371f"):
        assert lit in sh, lit


def test_thread_groups_and_shape_of_the_shipped_shaders():
    tg = {k: v["thread_group"] for k, v in FIX.items()}
    assert tg["CSAdvect"] == [8, 8, 1] and tg["CSProject3D"] == [4, 4, 4] and tg["CSProject2D"] == [8, 8, 1]
    assert tg["CSRayMarch"] == [8, 8, 1] and tg["CSRayMarchL"] == [4, 4, 4] and tg["CSSHCubeMap"] == [32, 1, 1]
    # structure the oracle relies on: one relaxation loop with 6 (3-D) / 4 (2-D) coherent neighbour loads + self
    assert FIX["CSProject3D"]["opcodes"]["LOOP"] == 1 and FIX["CSProject3D"]["opcodes"]["LD_UAV_TYPED"] == 7 + 6
    assert FIX["CSProject2D"]["opcodes"]["LD_UAV_TYPED"] == 5 + 4
    assert FIX["CSAdvect"]["opcodes"]["SAMPLE_L"] == 2 and FIX["CSAdvect"]["opcodes"]["EXP"] == 1
    assert FIX["CSRayMarchV"]["opcodes"]["LOOP"] == 1          # view march only
    assert FIX["CSRayMarch"]["opcodes"]["LOOP"] == 3           # view march + light march + AO march
    assert FIX["CSRayMarchL"]["opcodes"]["LOOP"] == 2
    assert FIX["CSSHSum"]["opcodes"]["SYNC"] == 5              # 32-lane LDS tree: 16, 8, 4, 2 (+1) steps


@pytest.mark.skipif(not os.path.isdir("/root/reference/Bin"), reason="reference not mounted (GPU box)")
def test_golden_vectors_regenerate_from_the_mounted_binaries():
    """re-run two of the shipped shaders in the interpreter and compare with the committed golden vectors"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_dxbc_golden as mk
    g = np.load(os.path.join(ROOT, "tests", "golden", "dxbc_sim.npz"))
    dt = np.float32(2.0 / 16)
    vo, co = mk.run_advect(g["advect_3d_f16_mirror_vel_in"], g["advect_3d_f16_mirror_col_in"], dt, "MIRROR", "R16G16B16A16_FLOAT")
    assert np.array_equal(vo, g["advect_3d_f16_mirror_vel_out"]) and np.array_equal(co, g["advect_3d_f16_mirror_col_out"])
    v0, p, _ = mk.run_project(g["project_3d_f32_vel_in"], g["project_3d_f32_p_in"], dt, "R32G32B32A32_FLOAT")
    assert np.array_equal(v0, g["project_3d_f32_vel_out"]) and np.array_equal(p, g["project_3d_f32_p_out"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/Bin"), reason="reference not mounted (GPU box)")
def test_fixture_matches_the_mounted_reference_and_decoder_self_checks():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dxbc
    import make_cso_fixture as mk
    for s in mk.SHADERS:
        assert mk.digest(os.path.join(mk.REF_BIN, s + ".cso")) == FIX[s], s
    n = 0
    for fn in sorted(os.listdir("/root/reference/Bin")):
        if fn.endswith(".cso"):
            dxbc.decode(open(os.path.join("/root/reference/Bin", fn), "rb").read())   # asserts operand/length consistency
            n += 1
    assert n == 17        # 9 compute + 8 graphics blobs ship in Bin/
