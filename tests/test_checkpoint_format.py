"""The whole-grid state file (fx_checkpoint_save / _load) as numpy sees it: header layout, field order, round trip,
rejection of foreign or truncated files.  No GPU: the library side is covered by tests/test_gpu_slabs.py."""
import numpy as np
import pytest

import fluidx12_amd as fx


def test_round_trip_and_layout(tmp_path):
    rng = np.random.default_rng(3)
    X, Y, Z = 12, 8, 5
    vel = rng.standard_normal((3, Z, Y, X)).astype(np.float32)
    col = rng.random((Z, Y, X, 4)).astype(np.float32)
    p = rng.standard_normal((Z, Y, X)).astype(np.float32)
    path = tmp_path / "s.fxck"
    fx.write_checkpoint(path, vel, col, p, storage=1, steps=77)
    raw = path.read_bytes()
    assert len(raw) == 64 + 8 * X * Y * Z * 4 + 8 * Z
    assert np.array_equal(np.frombuffer(raw[-8 * Z:], "<u8"), np.full(Z, 78))         # every plane carries the save's identity: steps + 1
    assert raw[:8] == b"FXCKPT03"
    assert tuple(np.frombuffer(raw, "<u4", 4, 8)) == (X, Y, Z, 1) and int(np.frombuffer(raw, "<u8", 1, 24)[0]) == 77
    assert raw[32:64] == bytes(32)
    body = np.frombuffer(raw[:-8 * Z], np.float32, offset=64)
    assert np.array_equal(body[:vel.size], vel.ravel()) and np.array_equal(body[-p.size:], p.ravel())
    for mm in (False, True):
        d = fx.read_checkpoint(path, mmap=mm)
        assert d["grid"] == (X, Y, Z) and d["storage"] == 1 and d["steps"] == 77
        assert np.array_equal(d["velocity"], vel) and np.array_equal(d["color"], col) and np.array_equal(d["pressure"], p)
        assert d["complete"].all() and d["complete"].shape == (Z,)
    # a plane left over from another save is not complete
    stale = bytearray(raw)
    stale[-8:] = np.array([5], "<u8").tobytes()
    (tmp_path / "stale.fxck").write_bytes(bytes(stale))
    c = fx.read_checkpoint(tmp_path / "stale.fxck")["complete"]
    assert c[:-1].all() and not c[-1]


def test_foreign_and_truncated_files_are_refused(tmp_path):
    bad = tmp_path / "bad.fxck"
    bad.write_bytes(b"NOTACKPT" + bytes(120))
    with pytest.raises(ValueError):
        fx.read_checkpoint(bad)
    good = tmp_path / "g.fxck"
    fx.write_checkpoint(good, np.zeros((3, 2, 4, 4)), np.zeros((2, 4, 4, 4)), np.zeros((2, 4, 4)))
    cut = tmp_path / "cut.fxck"
    cut.write_bytes(good.read_bytes()[:-16])
    with pytest.raises(ValueError):
        fx.read_checkpoint(cut)
    with pytest.raises(ValueError):
        fx.write_checkpoint(tmp_path / "x", np.zeros((3, 2, 4, 4)), np.zeros((2, 4, 4, 4)), np.zeros((2, 4, 5)))
