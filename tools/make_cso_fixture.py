#!/usr/bin/env python3
"""Digest of the reference's shipped compute-shader binaries (Bin/*.cso) -> tests/golden/cso_constants.json.

For every hot-path shader: its opcode histogram, thread-group size and the sorted set of 32-bit immediate
operands that are not small integers (as hex bit patterns).  That is data extracted from the reference's own
binaries -- the folded constants and instruction mix the oracle restates -- not shader source or a listing.
Run in the authoring container (needs /root/reference); the GPU box only reads the committed JSON.
"""
import collections
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dxbc  # noqa: E402

REF_BIN = "/root/reference/Bin"
SHADERS = ["CSAdvect", "CSProject2D", "CSProject3D", "CSRayMarch", "CSRayMarchL", "CSRayMarchV", "CSSHCubeMap", "CSSHSum",
           "CSSHNormalize"]


def digest(path):
    blob = open(path, "rb").read()
    version, ins = dxbc.decode(blob)
    ops = collections.Counter(i.op for i in ins if not i.op.startswith("DCL") and i.op != "CUSTOMDATA")
    imms = set()

    def walk(o):
        if o.type == "l":
            for u in o.imm:
                if 0x10000 <= u <= 0xFFFF0000:
                    imms.add(u)
        for ix in o.indices:
            if isinstance(ix, tuple):
                walk(ix[1])
    group = None
    for i in ins:
        if i.op == "DCL_THREAD_GROUP":
            group = list(i.extra)
        for o in i.operands:
            walk(o)
    return {"bytes": len(blob), "shader_model": "cs_%d_%d" % (version >> 4 & 0xF, version & 0xF), "thread_group": group,
            "instructions": sum(ops.values()), "opcodes": dict(sorted(ops.items())),
            "float_immediates": ["0x%08x" % u for u in sorted(imms)]}


def main():
    out = {s: digest(os.path.join(REF_BIN, s + ".cso")) for s in SHADERS}
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cso_constants.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", dst)


if __name__ == "__main__":
    main()
