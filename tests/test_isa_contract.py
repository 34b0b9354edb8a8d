"""Contracts between hand-written `s_waitcnt` counts and the code the compiler generates around them, checked on the gfx950 ISA
(hipcc cross-compiles without a GPU).  ADVICE r2: k_advect_lds hands a ring slot over with `s_waitcnt vmcnt(4)` = "everything but
this step's four stores has landed", which silently breaks if the compiler ever emits another number of store instructions."""
import os
import re
import subprocess

import pytest

from fluidx12_amd import build as b


def device_isa(source, tmp_path):
    out = tmp_path / (source + ".s")
    cmd = [b.hipcc()] + b.FLAGS + b.EXTRA_FLAGS.get(source, []) + ["-x", "hip", "--cuda-device-only", "-S", os.path.join(b.CSRC, source), "-o", str(out)]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return out.read_text().splitlines()


def kernels(lines, prefix):
    """{mangled name: body lines} of the functions whose name contains `prefix`"""
    out, cur = {}, None
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1) if prefix in m.group(1) else None
            if cur:
                out[cur] = []
        elif cur:
            if ln.strip().startswith(".end_amdhsa_kernel") or ln.strip().startswith(".Lfunc_end"):
                cur = None
            else:
                out[cur].append(ln.strip())
    return out


def test_advect_lds_hands_over_behind_exactly_its_stores(tmp_path):
    ks = kernels(device_isa("fx_advect_lds.hip", tmp_path), "k_advect_lds")
    assert len(ks) >= 8                                         # <HALF> x <tile rows> x <DEFER>
    for name, body in ks.items():
        # <HALF, TY, DEFER, P2, ALPHA>: with ALPHA a fifth store (the render's alpha side volume) sits behind the fill and the wait says vmcnt(5)
        m = re.search(r"k_advect_ldsILb[01]ELi\d+ELb([01])ELb[01]ELb([01])E", name)
        assert m, name
        deferring, n_st = m.group(1) == "1", 5 if m.group(2) == "1" else 4
        waits = [i for i, ln in enumerate(body) if ln.startswith("s_waitcnt vmcnt(%d)" % n_st) and body[i - 1].startswith(";;#ASMSTART")]   # the hand-written ones
        assert waits, name
        assert not [i for i, ln in enumerate(body) if ln.startswith("s_waitcnt vmcnt(%d)" % (9 - n_st)) and body[i - 1].startswith(";;#ASMSTART")], name
        for w in waits:
            stores = 0
            for ln in reversed(body[:w]):
                if ln.startswith(("global_load", "buffer_load", "flat_load", "s_barrier", "scratch_")):
                    break
                if ln.startswith(("global_store", "buffer_store", "flat_store")):
                    stores += 1
            # DEFER kernels note a far-tracing voxel with one more (older, lane-masked) store somewhere in front of the four (the compiler
            # rotates the loop, so it may sit behind the loop head): the wait then also covers that store -- never fewer operations
            # than the LDS-DMA loads
            assert stores in ((n_st, n_st + 1) if deferring else (n_st,)), (name, w, stores)
        # the LDS-DMA statements change SCC (s_add_u32 m0): they must say so
        assert any("global_load_lds" in ln for ln in body)
    src = open(os.path.join(b.CSRC, "fx_advect_lds.hip")).read()
    assert src.count('"scc"') >= 2 and "s_add_u32 m0" in src


def test_no_scratch_in_the_hot_kernels(tmp_path):
    """register-resident kernels must not spill (a spill turns a bandwidth-bound kernel into a scratch-bound one silently)"""
    for source, prefix in (("fx_jacobi_freeze.hip", "k_freeze_"), ("fx_advect_lds.hip", "k_advect_lds"), ("fx_advect_lds.hip", "k_advect_far"), ("fx_jacobi_stripm.hip", "k_freeze_strip3"),
                           ("fx_render_accel.hip", "k_view_slots"), ("fx_render_accel.hip", "k_light_rays"), ("fx_render_accel.hip", "k_build_fill"),
                           ("fx_render_accel.hip", "k_direct_march"), ("fx_jacobi_strip4.hip", "k_jacobi_strip4"), ("fx_jacobi_strip4.hip", "k_freeze_strip4"), ("fx_jacobi_strip3.hip", "k_jacobi_strip3c")):
        text = "\n".join(device_isa(source, tmp_path))
        for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
            if prefix in m.group(1) and "k_jacobi_strip4x" not in m.group(1) and "k_jacobi_strip4t" not in m.group(1):      # (the half-row and the tiled octet: their own test below)
                assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", m.group(2)), m.group(1)


@pytest.mark.parametrize("kernel,loops", [("k_jacobi_strip4x", 6), ("k_jacobi_strip4t", 3)])
def test_the_half_row_and_the_tiled_octet_keep_their_hot_loops_free_of_scratch(tmp_path, kernel, loops):
    """k_jacobi_strip4x (X = 512) carries four role bodies at 256 registers each, k_jacobi_strip4t (any other X > 256) three inside a loop
    over the pieces of a run; the allocator parks a few values of their PROLOGUES in scratch (< 256 bytes).  What must hold: two waves per
    SIMD (<= 256 registers), the LDS under 160 KiB, and no scratch instruction inside any of the steady-state loops (three z steps each: 3
    or 6 output rows stored per trip)"""
    lines = device_isa("fx_jacobi_strip4.hip", tmp_path)
    text = "\n".join(lines)
    m = [m for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S) if kernel in m.group(1)]
    assert len(m) == 2                                               # <NT = false / true>: plain and non-temporal output stores
    for mm in m:
        body = mm.group(2)
        assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1)) <= 256
        assert int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", body).group(1)) <= 160 * 1024
        assert int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1)) <= 256
    ks = kernels(lines, kernel)
    assert len(ks) == 2
    for code in ks.values():
        labels = {mm.group(1): i for i, ln in enumerate(code) for mm in [re.match(r"^(\.LBB\d+_\d+):", ln)] if mm}
        hot = []
        for i, ln in enumerate(code):
            mm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
                seg = [x for x in code[labels[mm.group(1)]:i + 1] if x and not x.startswith((".", ";"))]
                stores = sum(1 for x in seg if x.startswith(("global_store_dwordx4", "buffer_store_dwordx4")))
                if stores in (3, 6) and 900 < len(seg) < 1800:
                    hot.append(seg)
        assert len(hot) >= loops, len(hot)                           # (each of the role bodies' loops, seen through one or more back edges)
        for seg in hot:
            assert not [x for x in seg if x.startswith("scratch_")]
