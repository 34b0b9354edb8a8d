"""Reader / writer of the whole-grid state files of fx_checkpoint_save (fluidx12_amd/csrc/fx_checkpoint.cpp) in numpy -- no GPU
and no library needed: fixtures can be made or inspected anywhere.  Layout: 64-byte header ("FXCKPT03", X, Y, Z, storage, u64
steps, 32 reserved bytes), then float32 velocity[3][Z][Y][X], colour[Z][Y][X][4], pressure[Z][Y][X], then u64 mark[Z]: steps + 1
of the save that wrote that z plane completely (a writer marks its planes last; a plane is complete iff its mark names the
header's save), 0 = incomplete."""
import numpy as np

MAGIC = b"FXCKPT03"
HEADER = np.dtype([("magic", "S8"), ("X", "<u4"), ("Y", "<u4"), ("Z", "<u4"), ("storage", "<u4"), ("steps", "<u8"), ("reserved", "<u4", 8)])
assert HEADER.itemsize == 64


def read_checkpoint(path, mmap=False):
    """-> dict(grid=(X, Y, Z), storage, steps, velocity[3][Z][Y][X], color[Z][Y][X][4], pressure[Z][Y][X], complete[Z])"""
    h = np.fromfile(path, HEADER, 1)
    if h.size != 1 or bytes(h["magic"][0]) != MAGIC:
        raise ValueError("%s is not a FXCKPT03 file" % path)
    X, Y, Z = int(h["X"][0]), int(h["Y"][0]), int(h["Z"][0])
    n = X * Y * Z
    raw = np.memmap(path, np.uint8, "r", 64) if mmap else np.fromfile(path, np.uint8, offset=64)
    if raw.size != 8 * n * 4 + 8 * Z:
        raise ValueError("%s is truncated: %d payload bytes, expected %d" % (path, raw.size, 8 * n * 4 + 8 * Z))
    data = raw[:8 * n * 4].view(np.float32)
    return {"grid": (X, Y, Z), "storage": int(h["storage"][0]), "steps": int(h["steps"][0]),
            "velocity": data[:3 * n].reshape(3, Z, Y, X), "color": data[3 * n:7 * n].reshape(Z, Y, X, 4),
            "pressure": data[7 * n:].reshape(Z, Y, X), "marks": np.asarray(raw[8 * n * 4:]).view("<u8").copy(),
            "complete": np.asarray(raw[8 * n * 4:]).view("<u8") == int(h["steps"][0]) + 1}


def write_checkpoint(path, velocity, color, pressure, storage=0, steps=0):
    velocity, color, pressure = (np.ascontiguousarray(a, np.float32) for a in (velocity, color, pressure))
    Z, Y, X = pressure.shape
    if velocity.shape != (3, Z, Y, X) or color.shape != (Z, Y, X, 4):
        raise ValueError("shapes do not describe one grid")
    h = np.zeros(1, HEADER)
    h["magic"], h["X"], h["Y"], h["Z"], h["storage"], h["steps"] = MAGIC, X, Y, Z, storage, steps
    with open(path, "wb") as f:
        f.write(h.tobytes())
        for a in (velocity, color, pressure):
            f.write(a.tobytes())
        f.write(np.full(Z, steps + 1, "<u8").tobytes())
