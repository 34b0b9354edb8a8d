"""The one-process-per-GPU slab path (RcclTransport, fx_comm_init_rank, fx_comm_gather_color, bench.py under
torch.distributed.run) on a ONE-GPU box: the rank processes share GPU 0 and the product's RCCL calls are bound, through
FLUIDX_RCCL_LIB, to tests/mock_rccl (file rendezvous with RCCL's FIFO matching, byte-count checks and time-outs).
What this pins: every rank issues the same exchange sequence under every schedule (a disagreement is an error here,
a hang on real RCCL), and the result is bit-identical to the single-domain run.  It measures nothing."""
import json
import os
import shutil
import subprocess
import sys

import pytest

from test_dist_gloo import free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mock_lib():
    out = os.path.join(ROOT, "tests", "_build", "libmockrccl.so")
    src = os.path.join(ROOT, "tests", "mock_rccl", "mock_rccl.cpp")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.run(["g++", "-O1", "-shared", "-fPIC", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", out,
                        "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return out


def launch(nranks, script_args, mock_lib, tmp_path, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", FLUIDX_RCCL_LIB=mock_lib,
               FXMOCK_DIR=str(tmp_path), FXMOCK_TIMEOUT_S="120", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr",
           "127.0.0.1", "--master-port", str(free_port())] + script_args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    leftovers = [n for n in os.listdir(tmp_path)]
    shutil.rmtree(tmp_path, ignore_errors=True)
    return r, leftovers


@pytest.mark.parametrize("nranks,dims,halo_j,storage", [(2, "64x64x64", 8, "fp32"), (3, "64x64x96", 4, "fp32"),
                                                       (4, "64x64x128", 8, "fp16"),
                                                       # rows the four-sweep band kernels serve, thin slabs: x tiles, the octet's short z chunks, 512 as three tiles
                                                       (2, "320x320x40", 8, "fp32"), (3, "256x256x48", 5, "fp32"), (2, "512x512x34", 6, "fp32")])
def test_rank_processes_match_single_domain(nranks, dims, halo_j, storage, mock_lib, tmp_path):
    r, leftovers = launch(nranks, [os.path.join(ROOT, "tests", "mp_slab_worker.py"), dims, "6", "16", str(halo_j), storage],
                          mock_lib, tmp_path)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "OK %d ranks" % nranks in r.stdout
    assert leftovers == [], leftovers                    # every message consumed, every communicator directory removed


@pytest.mark.parametrize("nranks,steps", [(2, 3), (4, 3), (8, 9)])
def test_bench_under_torchrun_shared_gpu(nranks, steps, mock_lib, tmp_path):
    """(8 ranks, 9 steps: the driver's N = 8 launch shape -- (2G)^3 grid, 8-plane advection halo, marks on every fourth step)"""
    r, _ = launch(nranks, [os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--grid", "64", "--steps", str(steps), "--warmup", "1",
                           "--shared-gpu"], mock_lib, tmp_path)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == nranks and d["shared_gpu"] is True and d["value"] > 0
    assert d["config"]["schedule"]["picked"].startswith("fastest") and len(d["config"]["schedule"]["candidates"]) == 5
    assert "RCCL send/recv" in d["config"]["parallelism"] and d["roofline"]["achieved"] > 0
    sc = d["config"]["schedule"]
    assert all(c["exchange_calls_per_step"] > 0 and c["sent_MB_per_step"] > 0 for c in sc["candidates"])
    assert [c["overlap"] for c in sc["candidates"]] == [0, 0, 1, 2, 2]          # the plainest schedule first
    assert "link_model" in sc and sc["exchange_calls_per_step"] > 0
    assert r.stderr.count("bench.py candidate: {") == 5                      # one line per candidate as it completes
    # the line certifies what it timed: every rank's owned planes == the same planes of ONE domain stepped through the same frames
    assert d["multi_rank_parity"] == "bit-identical" and d["multi_rank_parity_detail"]["ranks"] == nranks
    assert d["multi_rank_parity_detail"]["steps_replayed"] == 5 * 4 + 1 + steps


def test_bench_notices_a_corrupted_halo(mock_lib, tmp_path):
    """bits of one halo message flip on their way to rank 1 (mock fault injection): the run still finishes and prints its line, but the
    line names the rank and the field, and the exit code is 4"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", FLUIDX_RCCL_LIB=mock_lib, FXMOCK_DIR=str(tmp_path), FXMOCK_TIMEOUT_S="120",
               HSA_ENABLE_IPC_MODE_LEGACY="0", FXMOCK_CORRUPT="1:40")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "64", "--steps", "3", "--warmup", "1", "--shared-gpu"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    shutil.rmtree(tmp_path, ignore_errors=True)
    assert "corrupted one byte" in r.stderr, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["multi_rank_parity"] != "bit-identical" and "differs from the single-domain replay" in d["multi_rank_parity"]
    assert r.returncode != 0


def test_mock_flags_a_byte_count_mismatch(mock_lib, tmp_path):
    """the checker checks: two ranks that disagree about a message size get an error, not a hang"""
    code = r'''
import ctypes as C, os, sys, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); dist.init_process_group("gloo", rank=rank, world_size=2)
m = C.CDLL(os.environ["FLUIDX_RCCL_LIB"]); hip = C.CDLL("libamdhip64.so")
uid = (C.c_char * 128)()
if rank == 0: m.ncclGetUniqueId(uid)
box = [bytes(uid)]; dist.broadcast_object_list(box, src=0); uid = (C.c_char * 128).from_buffer_copy(box[0])
class Id(C.Structure): _fields_ = [("b", C.c_char * 128)]
comm = C.c_void_p(); assert m.ncclCommInitRank(C.byref(comm), 2, Id(bytes(uid)), rank) == 0
t = torch.zeros(64, dtype=torch.uint8, device="cuda")
n = 64 if rank == 0 else 32
m.ncclGroupStart()
m.ncclSend(C.c_void_p(t.data_ptr()), C.c_size_t(n), 0, 1 - rank, comm, None)
m.ncclRecv(C.c_void_p(t.data_ptr()), C.c_size_t(n), 0, 1 - rank, comm, None)
rc = m.ncclGroupEnd()
dist.barrier(); m.ncclCommDestroy(comm); dist.destroy_process_group()
sys.exit(0 if rc != 0 else 3)
'''
    script = tmp_path.parent / "mock_mismatch.py"
    script.write_text(code)
    r, _ = launch(2, [str(script)], mock_lib, tmp_path, timeout=300)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])


def test_ranks_with_different_descriptors_are_refused_at_init(mock_lib, tmp_path):
    """fx_comm_init_rank compares a digest of the descriptor over the ranks (grid, halos, sweeps, mode, storage, addressing):
    a chain whose ranks would run different schedules fails on EVERY rank at set-up, before any exchange can mismatch"""
    code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import torch.distributed as dist
import fluidx12_amd as fx
rank = int(os.environ["RANK"]); dist.init_process_group("gloo", rank=rank, world_size=2)
f = fx.Fluid()
assert f.Init(64, 64, (32, 32, 32), slab=(rank * 16, 16), jacobi_iters=10 + rank, halo_advect=6, halo_jacobi=4, device=0)
box = [fx.comm_unique_id() if rank == 0 else None]; dist.broadcast_object_list(box, src=0)
try:
    f.comm_init_rank(box[0], rank, 2)
    refused = False
except fx.FluidxError as e:
    refused = e.status == -1
flags = [None, None]; dist.all_gather_object(flags, refused)
# the context is still a free one: a matching chain can be formed afterwards
g = fx.Fluid()
assert g.Init(64, 64, (32, 32, 32), slab=(rank * 16, 16), jacobi_iters=10, halo_advect=6, halo_jacobi=4, device=0)
box = [fx.comm_unique_id() if rank == 0 else None]; dist.broadcast_object_list(box, src=0)
g.comm_init_rank(box[0], rank, 2)
g.UpdateFrame(g.default_time_step(), 0); g.Simulate(0); g.Synchronize()
dist.barrier(); g.Release(); f.Release(); dist.barrier(); dist.destroy_process_group()
sys.exit(0 if all(flags) else 5)
'''
    script = tmp_path.parent / "mock_digest.py"
    script.write_text(code)
    r, leftovers = launch(2, [str(script)], mock_lib, tmp_path, timeout=300)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2500:])
    assert leftovers == [], leftovers


def test_one_rank_uploads_between_option_switches_and_the_chain_stays_in_step(mock_lib, tmp_path):
    """FX_OPT_ADAPTIVE_HALO off, a velocity upload into ONE rank, the option on again -- the workflow the header
    prescribes -- used to leave that rank exchanging halo_advect planes and its neighbour the measured need (mismatched send / recv
    counts: a hang on RCCL, an error on the mock).  Setting the option now drops the measurement on every rank.  Also: a checkpoint
    load made by every rank in the middle of a run (it used to be refused through fx_upload) resumes bit-identically."""
    code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch.distributed as dist
import fluidx12_amd as fx
from fluidx12_amd import capi
rank = int(os.environ["RANK"]); dist.init_process_group("gloo", rank=rank, world_size=2)
dims, ck = (32, 32, 64), sys.argv[1]
def chain():
    f = fx.Fluid()
    assert f.Init(64, 64, dims, slab=(rank * 32, 32), jacobi_iters=8, halo_advect=8, halo_jacobi=4, device=0)
    box = [fx.comm_unique_id() if rank == 0 else None]; dist.broadcast_object_list(box, src=0)
    f.comm_init_rank(box[0], rank, 2)
    return f
def steps(f, k0, n):
    for k in range(k0, k0 + n):
        f.UpdateFrame(np.float32(2.0 / 32), k % 3); f.Simulate(k % 3)
f = chain()
steps(f, 0, 4)                                            # the measured need is in use by now
f.Synchronize()
f.set_option(capi.OPT_ADAPTIVE_HALO, 0)                   # collective
if rank == 0:
    f.upload(fx.FIELD_VELOCITY, f.download(fx.FIELD_VELOCITY))    # one rank only (same values: the run stays comparable)
f.set_option(capi.OPT_ADAPTIVE_HALO, 1)                   # collective: every rank falls back to halo_advect planes for one step
steps(f, 4, 3)
f.Synchronize()
f.SaveCheckpoint(ck); dist.barrier()
steps(f, 7, 2); f.Synchronize()
after9 = f.download(fx.FIELD_COLOR)
f.LoadCheckpoint(ck); dist.barrier()                      # every rank, mid-run, adaptive halo on: refused before
steps(f, 7, 2); f.Synchronize()
same = bool((f.download(fx.FIELD_COLOR).view(np.uint8) == after9.view(np.uint8)).all())
# reference: a chain that just steps
g = chain(); steps(g, 0, 9); g.Synchronize()
same = same and bool((g.download(fx.FIELD_COLOR).view(np.uint8) == after9.view(np.uint8)).all()) and bool(after9.any())
flags = [None, None]; dist.all_gather_object(flags, same)
dist.barrier(); f.Release(); g.Release(); dist.barrier(); dist.destroy_process_group()
sys.exit(0 if all(flags) else 7)
'''
    script = tmp_path.parent / "mock_one_rank_upload.py"
    script.write_text(code)
    ck = tmp_path.parent / "mock_chain.fxck"
    r, _ = launch(2, [str(script), str(ck)], mock_lib, tmp_path, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])


def test_a_stalled_peer_ends_the_bench_with_an_error_instead_of_a_hang(mock_lib, tmp_path):
    """`python bench.py --gpus 2` as the driver starts it (the launcher path).  Rank 1 never enters the second
    schedule candidate; rank 0 blocks in its exchange; its watchdog names the phase and exits, torch.distributed.run tears the
    job down and the launcher returns non-zero -- well inside the launcher's own budget.  The candidate that completed before is on
    stderr."""
    import time
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), OMP_NUM_THREADS="1", FLUIDX_RCCL_LIB=mock_lib,
               FXMOCK_DIR=str(tmp_path), FXMOCK_TIMEOUT_S="600", HSA_ENABLE_IPC_MODE_LEGACY="0",
               FLUIDX_BENCH_FAULT="stall:1", FLUIDX_BENCH_WATCHDOG_S="8", FLUIDX_BENCH_TIMEOUT_S="150")
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "64", "--steps", "3", "--warmup", "1", "--shared-gpu"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    took = time.monotonic() - t0
    shutil.rmtree(tmp_path, ignore_errors=True)
    assert r.returncode != 0 and took < 140, (r.returncode, took, r.stderr[-2000:])
    assert "watchdog: rank 0 stuck in phase 'schedule candidate overlap=1" in r.stderr, r.stderr[-3000:]
    assert r.stderr.count("bench.py candidate: {") == 2 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("mode", ["async", "die"])
def test_a_failed_link_or_a_dead_peer_returns_comm_errors_instead_of_hanging(mode, mock_lib, tmp_path):
    """VERDICT r4 item 8.  Two rank processes on the mock; `async`: rank 1's ncclCommGetAsyncError turns bad after some exchanges
    (FXMOCK_ASYNC_ERROR) -- the product's poll in front of the next exchange sees it, aborts both communicators (ncclCommAbort) and
    returns FX_E_COMM; rank 0's wait ends on the abort marker.  `die`: rank 1 ends with os._exit in the middle of the run; rank 0's
    next wait (or its poll) notices the missing process.  In both cases every surviving rank reports FX_E_COMM (-5) within seconds -- the
    mock's own time-out is set to ten minutes --, the call after that fails at once too, and the process releases its context and exits."""
    import time
    env = dict(os.environ, OMP_NUM_THREADS="1", FLUIDX_RCCL_LIB=mock_lib, FXMOCK_DIR=str(tmp_path), FXMOCK_TIMEOUT_S="600",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    if mode == "async":
        env["FXMOCK_ASYNC_ERROR"] = "1:60"
    idf = str(tmp_path / "uid.bin")
    t0 = time.monotonic()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_fault_worker.py"), str(r), idf, mode], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=240))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()                                  # (exactly the children started above)
    took = time.monotonic() - t0
    shutil.rmtree(tmp_path, ignore_errors=True)
    assert took < 200, took
    lines = [l for o, _ in outs for l in o.splitlines() if l.startswith("RANK")]
    survivors = [0, 1] if mode == "async" else [0]
    assert len(lines) == len(survivors), (outs,)
    for l in lines:
        w = l.split()
        assert int(w[3]) == -5 and int(w[-1]) == -5, l           # FX_E_COMM, and again FX_E_COMM
        assert float(w[w.index("after") + 1]) < 60, l
    assert all(p.returncode == 0 for p in procs), [(p.returncode, e[-800:]) for p, (_, e) in zip(procs, outs)]
