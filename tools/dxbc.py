#!/usr/bin/env python3
"""DXBC (Shader Model 5.0) token-stream decoder.

Test/analysis infrastructure only -- never imported by the product path.

The reference ships its compute shaders pre-compiled as DXBC blobs
(/root/reference/Bin/*.cso).  They are the only *executable* statement of the
reference's arithmetic (there are no tests and the HLSL cannot be compiled in
this image), so this module decodes the SHEX chunk into a structured
instruction list that

  * `python tools/dxbc.py <file.cso>` prints as a listing (used to pin the
    operation order / folded constants the oracle restates), and
  * `tools/dxbc_interp.py` executes, to generate golden vectors from the
    reference's own shipped binaries (tests/golden/).

Format knowledge: the public d3d11TokenizedProgramFormat.hpp layout (opcode
token: [10:0] opcode, [30:24] length, [31] extended; operand token: [1:0]
component count, [3:2] selection mode, [11:4] mask/swizzle, [19:12] type,
[21:20] index dimension, [30:22] index representations, [31] extended).
Every instruction's operands must consume exactly the instruction's declared
length -- `decode()` asserts that, which is the decoder's self-check.
"""
import struct
import sys

OPCODES = """ADD AND BREAK BREAKC CALL CALLC CASE CONTINUE CONTINUEC CUT DEFAULT DERIV_RTX DERIV_RTY
DISCARD DIV DP2 DP3 DP4 ELSE EMIT EMITTHENCUT ENDIF ENDLOOP ENDSWITCH EQ EXP FRC FTOI FTOU GE IADD IF IEQ IGE ILT
IMAD IMAX IMIN IMUL INE INEG ISHL ISHR ITOF LABEL LD LD_MS LOG LOOP LT MAD MIN MAX CUSTOMDATA MOV MOVC MUL NE NOP
NOT OR RESINFO RET RETC ROUND_NE ROUND_NI ROUND_PI ROUND_Z RSQ SAMPLE SAMPLE_C SAMPLE_C_LZ SAMPLE_L SAMPLE_D
SAMPLE_B SQRT SWITCH SINCOS UDIV ULT UGE UMUL UMAD UMAX UMIN USHR UTOF XOR DCL_RESOURCE DCL_CONSTANT_BUFFER
DCL_SAMPLER DCL_INDEX_RANGE DCL_GS_OUTPUT_PRIMITIVE_TOPOLOGY DCL_GS_INPUT_PRIMITIVE DCL_MAX_OUTPUT_VERTEX_COUNT
DCL_INPUT DCL_INPUT_SGV DCL_INPUT_SIV DCL_INPUT_PS DCL_INPUT_PS_SGV DCL_INPUT_PS_SIV DCL_OUTPUT DCL_OUTPUT_SGV
DCL_OUTPUT_SIV DCL_TEMPS DCL_INDEXABLE_TEMP DCL_GLOBAL_FLAGS RESERVED0 LOD GATHER4 SAMPLE_POS SAMPLE_INFO RESERVED1
HS_DECLS HS_CONTROL_POINT_PHASE HS_FORK_PHASE HS_JOIN_PHASE EMIT_STREAM CUT_STREAM EMITTHENCUT_STREAM
INTERFACE_CALL BUFINFO DERIV_RTX_COARSE DERIV_RTX_FINE DERIV_RTY_COARSE DERIV_RTY_FINE GATHER4_C GATHER4_PO
GATHER4_PO_C RCP F32TOF16 F16TOF32 UADDC USUBB COUNTBITS FIRSTBIT_HI FIRSTBIT_LO FIRSTBIT_SHI UBFE IBFE BFI BFREV
SWAPC DCL_STREAM DCL_FUNCTION_BODY DCL_FUNCTION_TABLE DCL_INTERFACE DCL_INPUT_CONTROL_POINT_COUNT
DCL_OUTPUT_CONTROL_POINT_COUNT DCL_TESS_DOMAIN DCL_TESS_PARTITIONING DCL_TESS_OUTPUT_PRIMITIVE
DCL_HS_MAX_TESSFACTOR DCL_HS_FORK_PHASE_INSTANCE_COUNT DCL_HS_JOIN_PHASE_INSTANCE_COUNT DCL_THREAD_GROUP
DCL_UNORDERED_ACCESS_VIEW_TYPED DCL_UNORDERED_ACCESS_VIEW_RAW DCL_UNORDERED_ACCESS_VIEW_STRUCTURED
DCL_THREAD_GROUP_SHARED_MEMORY_RAW DCL_THREAD_GROUP_SHARED_MEMORY_STRUCTURED DCL_RESOURCE_RAW
DCL_RESOURCE_STRUCTURED LD_UAV_TYPED STORE_UAV_TYPED LD_RAW STORE_RAW LD_STRUCTURED STORE_STRUCTURED ATOMIC_AND
ATOMIC_OR ATOMIC_XOR ATOMIC_CMP_STORE ATOMIC_IADD ATOMIC_IMAX ATOMIC_IMIN ATOMIC_UMAX ATOMIC_UMIN
IMM_ATOMIC_ALLOC IMM_ATOMIC_CONSUME IMM_ATOMIC_IADD IMM_ATOMIC_AND IMM_ATOMIC_OR IMM_ATOMIC_XOR IMM_ATOMIC_EXCH
IMM_ATOMIC_CMP_EXCH IMM_ATOMIC_IMAX IMM_ATOMIC_IMIN IMM_ATOMIC_UMAX IMM_ATOMIC_UMIN SYNC DADD DMAX DMIN DMUL DEQ
DGE DLT DNE DMOV DMOVC DTOF FTOD EVAL_SNAPPED EVAL_SAMPLE_INDEX EVAL_CENTROID DCL_GS_INSTANCE_COUNT ABORT
DEBUG_BREAK""".split()

OPERAND_TYPES = """r v o x l l64 s t cb icb label vPrim oDepth null rasterizer oMask stream function_body
function_table interface function_input function_output vOutputControlPointID vForkInstanceID vJoinInstanceID
vicp vocp vpc vDomain this u g vThreadID vThreadGroupID vThreadIDInGroup vCoverage vThreadIDInGroupFlattened
vGSInstanceID oDepthGE oDepthLE vCycleCounter""".split()

# declaration opcodes whose payload after the (optional) operand is raw dwords
_DCL_RAW_TAIL = {
    "DCL_RESOURCE", "DCL_CONSTANT_BUFFER", "DCL_SAMPLER", "DCL_INPUT", "DCL_INPUT_SGV", "DCL_INPUT_SIV",
    "DCL_INPUT_PS", "DCL_INPUT_PS_SGV", "DCL_INPUT_PS_SIV", "DCL_OUTPUT", "DCL_OUTPUT_SGV", "DCL_OUTPUT_SIV",
    "DCL_UNORDERED_ACCESS_VIEW_TYPED", "DCL_UNORDERED_ACCESS_VIEW_RAW", "DCL_UNORDERED_ACCESS_VIEW_STRUCTURED",
    "DCL_THREAD_GROUP_SHARED_MEMORY_RAW", "DCL_THREAD_GROUP_SHARED_MEMORY_STRUCTURED", "DCL_RESOURCE_RAW",
    "DCL_RESOURCE_STRUCTURED",
}
_DCL_NO_OPERAND = {"DCL_TEMPS", "DCL_INDEXABLE_TEMP", "DCL_GLOBAL_FLAGS", "DCL_THREAD_GROUP",
                   "DCL_MAX_OUTPUT_VERTEX_COUNT", "DCL_GS_INPUT_PRIMITIVE", "DCL_GS_OUTPUT_PRIMITIVE_TOPOLOGY",
                   "DCL_GS_INSTANCE_COUNT", "DCL_STREAM", "HS_DECLS"}


class Operand:
    __slots__ = ("type", "ncomp", "sel", "mask", "swizzle", "indices", "imm", "modifier", "minprec")

    def __init__(self):
        self.type = None
        self.ncomp = 0
        self.sel = None          # 'mask' | 'swizzle' | 'select1' | None
        self.mask = 0
        self.swizzle = (0, 1, 2, 3)
        self.indices = []        # each: int or (int, Operand) for relative
        self.imm = None          # tuple of raw uint32
        self.modifier = 0        # 0 none 1 neg 2 abs 3 -abs
        self.minprec = 0

    def __repr__(self):
        t = self.type
        if t in ("l",):
            vals = []
            for u in self.imm:
                f = struct.unpack("<f", struct.pack("<I", u))[0]
                # print small ints as ints, everything else as float + hex
                if u < 0x10000 or u > 0xFFFF0000:
                    vals.append("%d" % (u if u < 0x80000000 else u - (1 << 32)))
                else:
                    vals.append("%.9g[0x%08x]" % (f, u))
            return "l(" + ", ".join(vals) + ")"
        s = t
        for ix in self.indices:
            if isinstance(ix, tuple):
                s += "[%s + %d]" % (ix[1], ix[0]) if ix[0] else "[%s]" % (ix[1],)
            else:
                s += "[%d]" % ix if (t in ("cb", "icb", "x") or len(self.indices) > 1) else "%d" % ix
        if self.ncomp == 4:
            if self.sel == "mask":
                s += "." + "".join(c for i, c in enumerate("xyzw") if self.mask >> i & 1)
            elif self.sel == "swizzle":
                s += "." + "".join("xyzw"[c] for c in self.swizzle)
            elif self.sel == "select1":
                s += "." + "xyzw"[self.swizzle[0]]
        if self.modifier == 1:
            s = "-" + s
        elif self.modifier == 2:
            s = "|" + s + "|"
        elif self.modifier == 3:
            s = "-|" + s + "|"
        return s


class Instr:
    __slots__ = ("op", "ctrl", "operands", "raw", "extended", "offsets", "sat", "test_nz", "extra")

    def __init__(self):
        self.op = None
        self.ctrl = 0
        self.operands = []
        self.raw = ()
        self.extended = []
        self.offsets = (0, 0, 0)   # sample/ld immediate texel offsets
        self.sat = False
        self.test_nz = False
        self.extra = ()

    def __repr__(self):
        name = self.op.lower()
        if self.sat:
            name += "_sat"
        if self.op in ("IF", "BREAKC", "CONTINUEC", "RETC", "DISCARD", "CALLC"):
            name += "_nz" if self.test_nz else "_z"
        if any(self.offsets):
            name += "_aoffimmi(%d,%d,%d)" % self.offsets
        s = name + " " + ", ".join(repr(o) for o in self.operands)
        if self.extra:
            s += "  ; " + " ".join("%#x" % x for x in self.extra)
        return s


def _sext4(v):
    return v - 16 if v & 8 else v


def _parse_operand(tok, pos):
    o = Operand()
    t0 = tok[pos]
    pos += 1
    nc = t0 & 3
    o.ncomp = {0: 0, 1: 1, 2: 4, 3: -1}[nc]
    if nc == 2:
        mode = (t0 >> 2) & 3
        if mode == 0:
            o.sel = "mask"
            o.mask = (t0 >> 4) & 0xF
        elif mode == 1:
            o.sel = "swizzle"
            o.swizzle = tuple((t0 >> (4 + 2 * i)) & 3 for i in range(4))
        else:
            o.sel = "select1"
            o.swizzle = ((t0 >> 4) & 3,) * 4
    ty = (t0 >> 12) & 0xFF
    o.type = OPERAND_TYPES[ty]
    idim = (t0 >> 20) & 3
    reps = [(t0 >> (22 + 3 * i)) & 7 for i in range(3)]
    if t0 >> 31:
        ext = tok[pos]
        pos += 1
        if (ext & 0x3F) == 1:
            o.modifier = (ext >> 6) & 0xFF
            o.minprec = (ext >> 14) & 7
        assert not ext >> 31
    if o.type == "l":
        n = 4 if nc == 2 else 1
        o.imm = tuple(tok[pos:pos + n])
        pos += n
    elif o.type == "l64":
        n = 8 if nc == 2 else 2
        o.imm = tuple(tok[pos:pos + n])
        pos += n
    for i in range(idim):
        r = reps[i]
        if r == 0:
            o.indices.append(tok[pos])
            pos += 1
        elif r == 2:
            rel, pos = _parse_operand(tok, pos)
            o.indices.append((0, rel))
        elif r == 3:
            imm = tok[pos]
            pos += 1
            rel, pos = _parse_operand(tok, pos)
            o.indices.append((imm, rel))
        else:
            raise NotImplementedError("index representation %d" % r)
    return o, pos


def chunks(blob):
    assert blob[:4] == b"DXBC", "not a DXBC container"
    n = struct.unpack_from("<I", blob, 28)[0]
    out = {}
    for off in struct.unpack_from("<%dI" % n, blob, 32):
        tag = blob[off:off + 4].decode()
        size = struct.unpack_from("<I", blob, off + 4)[0]
        out[tag] = blob[off + 8:off + 8 + size]
    return out


def decode(blob):
    """Return (version, [Instr]) for the SHEX/SHDR chunk of a DXBC blob."""
    ch = chunks(blob)
    code = ch.get("SHEX") or ch.get("SHDR")
    tok = struct.unpack("<%dI" % (len(code) // 4), code)
    version, length = tok[0], tok[1]
    assert length == len(tok), (length, len(tok))
    pos = 2
    out = []
    while pos < length:
        t0 = tok[pos]
        opc = t0 & 0x7FF
        ins = Instr()
        ins.op = OPCODES[opc]
        ins.ctrl = (t0 >> 11) & 0x1FFF
        if ins.op == "CUSTOMDATA":
            n = tok[pos + 1]
            ins.raw = tok[pos + 2:pos + n]
            ins.ctrl = t0 >> 11
            out.append(ins)
            pos += n
            continue
        ilen = (t0 >> 24) & 0x7F
        end = pos + ilen
        ins.sat = bool(t0 >> 13 & 1)
        ins.test_nz = bool(t0 >> 18 & 1)
        p = pos + 1
        ext = t0 >> 31
        while ext:
            e = tok[p]
            p += 1
            ins.extended.append(e)
            if (e & 0x3F) == 1:   # sample controls: immediate texel offsets (4-bit signed)
                ins.offsets = (_sext4(e >> 9 & 0xF), _sext4(e >> 13 & 0xF), _sext4(e >> 17 & 0xF))
            ext = e >> 31
        if ins.op in _DCL_NO_OPERAND:
            ins.extra = tok[p:end]
            p = end
        else:
            if ins.op in _DCL_RAW_TAIL:
                o, p = _parse_operand(tok, p)
                ins.operands.append(o)
                ins.extra = tok[p:end]
                p = end
            else:
                while p < end:
                    o, p = _parse_operand(tok, p)
                    ins.operands.append(o)
        assert p == end, "operand decode overran instruction %s (%d != %d)" % (ins.op, p, end)
        out.append(ins)
        pos = end
    assert pos == length
    return version, out


def listing(blob):
    version, ins = decode(blob)
    kind = {0: "ps", 1: "vs", 2: "gs", 3: "hs", 4: "ds", 5: "cs"}[version >> 16]
    lines = ["%s_%d_%d" % (kind, version >> 4 & 0xF, version & 0xF)]
    depth = 0
    for i in ins:
        if i.op in ("ENDIF", "ENDLOOP", "ELSE", "ENDSWITCH"):
            depth -= 1
        if i.op == "CUSTOMDATA":
            lines.append("  " * depth + "customdata[%d dwords, class %d]" % (len(i.raw), i.ctrl & 0x1FFFFF))
        else:
            lines.append("  " * depth + repr(i))
        if i.op in ("IF", "LOOP", "ELSE", "SWITCH"):
            depth += 1
    return "\n".join(lines)


if __name__ == "__main__":
    for fn in sys.argv[1:]:
        print("//", fn)
        print(listing(open(fn, "rb").read()))
