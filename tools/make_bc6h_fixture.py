#!/usr/bin/env python3
"""tests/golden/bc6h_fixture.npz: data for the BC6H / DDS tests that must run without /root/reference.

  * dds_mip3:   a complete little DDS cube map (DX10 header + mip 3 of every face of the reference's
                Bin/Assets/rnl_cross.dds: 6 x 8 x 8 blocks = 6 KiB of the asset's DATA), so that the container parser and the
                decoder see real encoder output in 11 of the 14 modes
  * cube_mip3:  what the oracle decodes from it, float32 [6][32][32][3]
  * down_mip2:  the 2 x 2 box filter of the oracle's decode of mip 2 -- the independent witness: mip 3 was encoded from
                (about) this, in other modes and partitions, so decode(mip 3) must match it to BC6H quantisation
  * sh_mip3:    the oracle's SH projection of cube_mip3 (9 x 3)
  * mode_hist:  blocks per mode over the whole asset (mip 0..3), for the record

Only runs in the authoring container.  Nothing here touches the HIP library.
"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

SRC = "/root/reference/Bin/Assets/rnl_cross.dds"


def main():
    d = open(SRC, "rb").read()
    hdr = bytearray(d[:148])
    width = struct.unpack_from("<I", hdr, 16)[0]
    mips = struct.unpack_from("<I", hdr, 28)[0]
    sizes = [max(1, (width >> m) // 4) ** 2 * 16 for m in range(mips)]
    per_face = sum(sizes)
    mip = 3
    n = width >> mip
    struct.pack_into("<I", hdr, 12, n)            # height
    struct.pack_into("<I", hdr, 16, n)            # width
    struct.pack_into("<I", hdr, 20, max(1, n // 4) * 16 * max(1, n // 4))   # linear size of the top mip
    struct.pack_into("<I", hdr, 28, 1)            # one mip
    body = b"".join(d[148 + f * per_face + sum(sizes[:mip]):148 + f * per_face + sum(sizes[:mip + 1])] for f in range(6))
    small = bytes(hdr) + body
    cube3, h3 = orc.dds_bc6h_cube(small, 0)
    cube2, _ = orc.dds_bc6h_cube(d, 2)
    down = cube2.reshape(6, n, 2, n, 2, 3).mean(axis=(2, 4)).astype(np.float32)
    hist = sum(orc.dds_bc6h_cube(d, m)[1] for m in range(4))
    out = {"dds_mip3": np.frombuffer(small, np.uint8), "cube_mip3": cube3, "down_mip2": down,
           "sh_mip3": orc.sh_transform(cube3), "mode_hist": hist}
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "bc6h_fixture.npz"), **out)
    rel = np.abs(cube3 - down) / (np.abs(down) + 0.05)
    print("fixture: %d bytes of DDS, modes in mip 3 %s, median rel. distance to the box-filtered mip 2: %.4f" % (
        len(small), h3.tolist(), float(np.median(rel))))


if __name__ == "__main__":
    main()
