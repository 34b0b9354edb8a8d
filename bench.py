#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (driver contract, see BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the simulation hot path -- Fluid::UpdateFrame + Fluid::Simulate = semi-Lagrangian
advection, divergence, 40 lock-step Jacobi sweeps, projection -- over the 256^3 fp32 smoke grid of
BASELINE.json configs[2] (zero-initialised fields, built-in Gaussian impulse, dt = 2/Y, CLAMP sampler),
with every field resident in HBM before the timed region.  metric = voxel-updates/s.

N > 1 (launched by torch.distributed.run, one rank per GPU): the grid is cut into N z-slabs, one per rank, neighbour
halo planes exchanged with RCCL send/recv pairs over xGMI inside libfluidx_hip.so.  Default = weak scaling: 16.8 M
voxels per GPU (256^3, 256x256x512, 256x256x1024, 512^3 for N = 1, 2, 4, 8 -- see workload_grid); --scaling strong
keeps 256^3 for every N. torch.distributed only carries the rendezvous (RCCL unique id), the barriers and
the max-over-ranks reduction of the step time.

The JSON line carries `roofline` (dominant kernel = the Jacobi sweep: 12 algorithmic bytes per cell-sweep,
timed with HIP events on the kernel's own stream inside the timed region) and `cpu_baseline` (the CPU
oracle -- a scalar C++ port of the reference shaders, OpenMP over planes -- timed on the host cores on a
bounded sample of the same workload).  The oracle is used ONLY for that reported baseline.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
JACOBI_BYTES_PER_CELL_SWEEP = 12.0   # read p, read b, write p'  


def step_bytes_per_voxel(iters, storage):
    V, Cb = (12, 16) if storage == "fp32" else (6, 8)
    return 5 * V + 2 * Cb + 2 * 4 + 12 * iters            # 5V + 2C + 2S + 12 N


def _source_hash(kernel):
    from fluidx12_amd.build import kernel_source_hash
    return kernel_source_hash(kernel)


def pmc_traffic(kernel, grid, iters, storage, mode="fixed"):
    """HBM-side bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary (tools/pmc_summary.py) of this
    workload: (bytes, file, stale).  bench.py cannot run the profiler on itself; the summary is produced by the same command under
    rocprofv3 --pmc and committed under profiles/, stamped per kernel with a hash of the kernel's source file + build flags
    (fluidx12_amd.build.kernel_source_hash).  A summary whose stamp differs from the tree's -- the kernel changed and was not
    re-profiled -- is reported as stale and its figure is NOT used.  None when no summary matches the workload at all."""
    import glob
    best = None
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json"))):
        try:
            d = json.load(open(fn))
        except Exception:
            continue
        if (d.get("grid"), d.get("iters"), d.get("storage"), d.get("mode", "fixed")) == (grid, iters, storage, mode) and kernel in d.get("kernels", {}):
            e = d["kernels"][kernel]
            best = (e["traffic"], os.path.basename(fn), e.get("source_hash") != _source_hash(kernel))
    return best


def baseline_config_label(GX, GY, GZ, iters, storage, N):
    """which BASELINE.json config this line is (1-based as listed there), or how it differs from the nearest one"""
    for c, (g, it, st) in BASELINE_CONFIGS.items():
        if (GX, GY, GZ) == (g, g, g) and iters == it and storage == st:
            ranks = {4: 8}.get(c, 1)
            if N == ranks:
                return "BASELINE configs[%d]%s" % (c - 1, " (simulation half; the ray march is reported under `render`)" if c == 3 else "")
            return "BASELINE configs[%d]'s grid and sweep count on %d GPU%s instead of %d" % (c - 1, N, "s" if N > 1 else "", ranks)
    if (GX, GY, GZ) == (512, 512, 512) and storage == "fp32":
        return "BASELINE configs[3]'s 512^3 grid with %d instead of 80 sweeps (the weak-scaling series keeps the single-GPU sweep count; --config 4 runs configs[3] itself)" % iters
    return "not a BASELINE config (weak-scaling stack of the 256^3 problem or a custom grid)"


def limiter_note(kernel):
    """what the newest committed SQ-counter summary (tools/sq_summary.py -> profiles/r*_sq_counters.json) taken on THIS kernel source
    says binds `kernel` (summaries of an older version of the kernel are skipped)"""
    import glob
    note = None
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sq_counters*.json"))):
        try:
            d = json.load(open(fn))
        except Exception:
            continue
        k = d.get("kernels", {}).get(kernel)
        if k and k.get("limiter") and k.get("source_hash") == _source_hash(kernel):
            note = "%s (%s)" % (k["limiter"], os.path.basename(fn))
    return note


def step_traffic(grid, iters, storage, mode="fixed"):
    """fabric bytes ONE step moves, summed over its kernels from the newest committed PMC summary of this workload (the summary was
    taken with `bench.py --steps 4 --warmup 1`: every kernel's dispatch count / 5 = launches per step): (bytes, file, stale kernels).
    None if no summary matches."""
    import glob
    best = None
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json"))):
        try:
            d = json.load(open(fn))
        except Exception:
            continue
        if (d.get("grid"), d.get("iters"), d.get("storage"), d.get("mode", "fixed")) == (grid, iters, storage, mode) and d.get("kernels"):
            stale = sorted(k for k, e in d["kernels"].items() if e.get("source_hash") != _source_hash(k))
            best = (sum(k["traffic"] * k["dispatches"] / float(d.get("steps_profiled", 5)) for k in d["kernels"].values()), os.path.basename(fn), stale)
    return best


def render_pmc(grid, storage, has_sh):
    """cache hit rates of the render kernels from the newest committed counter summary of this workload (tools/render_pmc_summary.py
    -> profiles/r*_render_pmc*.json; TCC = L2, TCP = the CUs' vector L1), per kernel, with a `stale` flag when the kernel source changed
    since (fluidx12_amd.build.kernel_source_hash).  None when there is no summary."""
    import glob
    best = None
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_render_pmc*.json"))):
        try:
            d = json.load(open(fn))
        except Exception:
            continue
        if (d.get("grid"), d.get("storage"), bool(d.get("has_sh"))) == (grid, storage, bool(has_sh)) and d.get("kernels"):
            best = {"source": os.path.basename(fn), "frame": d.get("frame"),
                    "kernels": {k: {"l2_hit_rate": e.get("l2_hit_rate"), "l1_hit_rate": e.get("l1_hit_rate"), "avg_us": e.get("avg_us"),
                                    "stale": e.get("source_hash") != _source_hash(k)} for k, e in d["kernels"].items()}}
    return best


def workload_grid(G, N, scaling):
    """Grid of the N-rank run.  strong: the same G^3 for every N.  weak (default): 16.8 M voxels per GPU at G = 256.  N = 2 and 4
    stack G^3 blocks along z (G x G x 2G, G x G x 4G): every rank owns exactly the single-GPU problem, the textbook weak-scaling
    set-up, and a halo plane is G^2 cells.  N = 8 is (2G)^3 = BASELINE configs[3]'s 512^3 on 8 GPUs (64-plane slabs of 512^2); other N
    stack along z.  Returns ((X, Y, Z), advect halo planes): the halo covers the z back-trace reach measured with
    tools/reach_probe.py over 600 steps (11.9 / 18.8 / 5.1 cells for N = 2 / 4 / 8; a reach r needs floor(r) + 2 planes) + margin; a
    longer run that outgrows it stops with FX_E_HALO instead of computing something else.  (Until late in round 1 N = 4 ran
    2G x 2G x G: 42 + 40 MB per face and step instead of 39 + 10, and 1.48 instead of 1.14 ms of kernels per rank.)"""
    if N == 1:
        return (G, G, G), 0
    if scaling == "strong":
        return (G, G, G), (8 if G >= 512 else 0)      # 512^3: measured z reach 5.1 cells over 600 steps; 0 = the library default (6)
    if scaling == "weak256":
        # ONE kernel family over the whole table: G x G x (G N) for every N (the default table's N = 8 point is (2G)^3 -- other row length,
        # other kernels: a kink there is the kernels', not the links').  Halo: the z reach grows with the depth (dt = 2 / Y, reach in cells
        # ~ Z): 11.9 / 18.8 cells measured at N = 2 / 4 (tools/reach_probe.py), twice the N = 4 figure assumed at N = 8; FX_E_HALO if wrong.
        return (G, G, G * N), {2: 16, 4: 22, 8: 44}.get(N, 6 * N + 2)
    table = {2: ((G, G, 2 * G), 16), 4: ((G, G, 4 * G), 22), 8: ((2 * G, 2 * G, 2 * G), 8)}
    return table.get(N, ((G, G, G * N), 6 * N + 2))


# (FX_OPT_OVERLAP, FX_OPT_JACOBI_ROUND): advection halo behind the interior advection + serial pressure rounds of 8 sweeps;
# pressure exchanges behind the interior sweeps of rounds of 8 (also with the next step's colour halo sent behind the pressure
# phase) and of 4 sweeps (face planes first); nothing overlapped
# Order: the plainest schedule first (everything on the compute stream), then one more overlapped piece at a time -- if the first
# hardware run dies in one of them, the lines already printed say how far it got.  (0, 4) is there for the link model: two serial
# schedules with different message counts separate per-call latency from bandwidth.
SCHEDULE_CANDIDATES = [(0, 8), (0, 4), (1, 8), (2, 8), (2, 4)]
# FX_OPT_OVERLAP 3 drives a second RCCL communicator concurrently with the first; it has only ever run against the mock and the
# loop-back transport, so it is a candidate only on request (FLUIDX_BENCH_OVERLAP3=1 or --schedule 3,8)
if os.environ.get("FLUIDX_BENCH_OVERLAP3", "0") == "1":
    SCHEDULE_CANDIDATES.insert(2, (3, 8))

# BASELINE.json configs[i] -> (grid, sweeps, storage); configs[0] (64^2 2-D, CPU only) is a parity-test case, not a bench line
BASELINE_CONFIGS = {2: (128, 40, "fp32"), 3: (256, 40, "fp32"), 4: (512, 80, "fp32"), 5: (256, 40, "fp16")}


def pressure_round(GX, nz_per_rank, iters):
    """sweeps per pressure exchange: the library default of 8 = 4 + 4 sweeps (k_jacobi_strip4o / k_jacobi_strip4x).  With the four-sweep
    kernels switched off (FLUIDX_JACOBI_PREFER4=0), slab ranks thick enough for the three-sweep kernels (X = 256 from 12.6 M cells,
    X = 512 from 16.8 M -- a 64-plane rank of BASELINE configs[3]) take rounds of 9 = 3 + 3 + 3 instead of 8 = 3 + 3 + 2."""
    three = (GX == 256 and GX * GX * nz_per_rank >= 3 << 22) or (GX == 512 and GX * GX * nz_per_rank >= 1 << 24)
    if os.environ.get("FLUIDX_JACOBI_PREFER4", "1") != "0":
        return 8                                 # round 6: both row lengths have four-sweep kernels from these sizes on: 8 = 4 + 4 (9 would be 4 + 3 + 2)
    return 9 if (three and iters >= 9) else 8


def slab_for_rank(Z, rank, world):
    z0 = rank * Z // world
    z1 = (rank + 1) * Z // world
    return z0, z1 - z0


def cpu_baseline(grid, iters, budget_s=20.0, mode=0, half=False, address=0):
    """Time the CPU oracle (port of the reference shaders) on a bounded sample of the same workload:
    full simulation steps on a grid x grid x nz sub-volume sized for ~budget_s of CPU work.  The thread count is the
    one that runs fastest on a thin calibration slab (the visible CPU count of a container can exceed what it may use)."""
    import ctypes
    from oracle import orc                       # cpu_baseline leg only
    ncores = os.cpu_count() or 1
    try:
        ncores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    orc.lib()
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
    except OSError:
        gomp = None

    def slab_rate(nz, steps, threads):
        if gomp is not None:
            gomp.omp_set_num_threads(int(threads))
        sim = orc.Sim(grid, grid, nz, iters=iters, mode=mode, address=address, half=half)
        sim.step(2.0 / grid)                     # warm-up (page faults, OpenMP pool)
        t0 = time.perf_counter()
        for _ in range(steps):
            sim.step(2.0 / grid)
        return float(grid) * grid * nz * steps / (time.perf_counter() - t0)

    cands = sorted({t for t in (4, 8, 16, 32, 64, 96, 128, 192, 256, ncores) if t <= ncores}) if gomp is not None else [ncores]
    tried = {}
    t_start = time.perf_counter()
    for t in cands:
        tried[t] = slab_rate(8, 1, t)
        if time.perf_counter() - t_start > 0.4 * budget_s:
            break
    threads = max(tried, key=tried.get)
    steps = 2
    # three timings of a third of the remaining budget each: the line states their median and their spread (the figure moved by 1.5 x
    # between boxes in round 4: a shared host, a thread count picked on a thin slab)
    nz = int(max(8, min(grid, 0.2 * budget_s * tried[threads] / (float(grid) * grid * (steps + 1)))))
    rates = []
    for _ in range(3):
        rates.append(slab_rate(nz, steps, threads))
        if time.perf_counter() - t_start > 1.5 * budget_s:
            break
    rate = sorted(rates)[len(rates) // 2]
    single = slab_rate(4, 1, 1) if gomp is not None else None       # the pure scalar replay, one thread, a 4-plane slab
    fmt = ("%d steps of advect+divergence+%d Jacobi+project on a %dx%dx%d sub-volume of the %d^3 workload "
           + ("(reference configuration: sweep cap + per-cell early-out, RGBA16F storage) " if mode else "")
           + "(oracle/liborc.so, -O3, OpenMP over planes; %d threads = the fastest of %s on a thin slab, %d CPUs visible)")
    return {"value": rate, "unit": "voxel-updates/s", "cores": threads, "kind": "port", "single_thread_value": single,
            "repeats": rates, "spread": [min(rates), max(rates)], "calibration_by_threads": {str(k): v for k, v in sorted(tried.items())},
            "sample": fmt % (steps, iters, grid, grid, nz, grid, threads, sorted(tried), ncores) + "; value = median of %d timings" % len(rates)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=32)         # 32 warm-up + 100 timed steps
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4, 5],
                    help="BASELINE.json config (1-based as listed there): 2 = 128^3/40, 3 = 256^3/40 (the default workload), "
                         "4 = 512^3/80 sweeps (on 8 ranks: 64-plane z-slabs), 5 = 256^3 fp16 storage; sets --grid/--iters/--storage")
    ap.add_argument("--grid", type=int, default=None)
    ap.add_argument("--iters", type=int, default=None)
    ap.add_argument("--storage", default=None, choices=["fp32", "fp16"])
    ap.add_argument("--mode", default="fixed", choices=["fixed", "faithful"],
                    help="pressure solve: `fixed` = --iters lock-step sweeps (BASELINE's 20 / 40 / 80); `faithful` = the reference's own loop "
                         "(CSPoisson.hlsli:8-26: at most --iters sweeps, default 64, a cell stops once a sweep moves it by < 1e-3)")
    ap.add_argument("--address", default="clamp", choices=["clamp", "mirror"], help="advection sampler (FluidEZ = clamp, Fluid = mirror)")
    ap.add_argument("--reference-config", action="store_true",
                    help="the configuration the reference itself runs (Fluid.cpp:207-221, CSProject3D.hlsl:13): --mode faithful --iters 64 --storage fp16")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong", "weak256"],
                    help="how the grid grows with --gpus (N > 1): weak = 16.8 M voxels per GPU (N = 8: the 512^3 of BASELINE configs[3]); weak256 = the "
                         "same on ONE kernel family, G x G x (G N) for every N; strong = the same G^3 for every N")
    ap.add_argument("--schedule", default="auto", help="N > 1: 'auto' times the slab schedules below for 3 steps each before the "
                    "warm-up and keeps the fastest, or 'OVERLAP,ROUND' (fx_set_option values) to pin one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-render", action="store_true", help="skip the (untimed-for-value) ray-march measurement")
    ap.add_argument("--no-developed", action="store_true", help="skip the second timing of the step at frame 132 (`developed_plume`)")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--loopback", type=int, default=0,
                    help="run the N-rank code path (grid, slabs, halos, schedule selection) with N slab contexts in THIS process on "
                         "one GPU through the loop-back transport: a functional check of the multi-rank path, never a scaling number")
    ap.add_argument("--group", default="shared", choices=["shared", "peer"],
                    help="--loopback: `shared` = every slab context on ONE compute stream (fx_comm_init_local); `peer` = every context on its "
                         "own streams, halo planes pulled out of the neighbour's memory (fx_comm_init_peer)")
    ap.add_argument("--peer-devices", action="store_true",
                    help="--loopback N --group peer: slab r lives on device r -- ONE process driving N GPUs, planes travelling by "
                         "hipMemcpyPeerAsync.  This is a measurement (n_gpus = N), unlike the one-GPU loop-back")
    ap.add_argument("--preheat", action="store_true",
                    help="run ~80 ms of the same step on a scratch context in front of the warm-up steps (`device_preheat` in the line).  OFF by "
                         "default since round 6: `value` is the contract's protocol and nothing else -- W warm-up steps, K timed steps; the figure on "
                         "a device that is already awake is reported beside it as `warm_device` (the same W + K steps on a second context)")
    ap.add_argument("--no-preheat", action="store_true", help="(accepted for the round-5 tools; the default now)")
    ap.add_argument("--no-warm-leg", action="store_true", help="skip the `warm_device` leg")
    ap.add_argument("--no-parity", action="store_true",
                    help="N > 1: skip the single-domain replay that certifies the timed steps (`multi_rank_parity`); for profiler passes only")
    ap.add_argument("--no-peer-leg", action="store_true",
                    help="--gpus N > 1: do not time the in-process peer transport (a child process of rank 0, after the RCCL measurement) "
                         "beside the RCCL line")
    ap.add_argument("--shared-gpu", action="store_true",
                    help="functional check only: all ranks of a torch.distributed.run launch use GPU 0 (torch.distributed over gloo; "
                         "the product's RCCL calls must be redirected with FLUIDX_RCCL_LIB=tests/_build/libmockrccl.so, because "
                         "RCCL refuses two ranks on one device); never a measurement, the JSON line says so")
    ap.add_argument("--dry-run", action="store_true",
                    help="distributed plumbing only (no GPU work); used by the gloo CPU tests, never a measurement")
    args = ap.parse_args()
    # (the CPU baseline's OpenMP threads are left to the scheduler: pinned with OMP_PROC_BIND=close / OMP_PLACES=cores the same sample ran at
    # 15.8 instead of 56 M voxel-updates/s on the box of round 6 -- the visible CPUs of a container are not its cores; unpinned the three
    # timings of a run agree to 3 %)
    cg, ci, cs = BASELINE_CONFIGS.get(args.config, (256, 40, "fp32"))
    if args.reference_config:
        args.mode = "faithful"
        cs = "fp16"
    if args.mode == "faithful":
        ci = 64                                    # CSProject3D.hlsl:13
    if args.grid is None:
        args.grid = cg
    if args.iters is None:
        args.iters = ci
    if args.storage is None:
        args.storage = cs
    if args.config == 4 and (args.gpus > 1 or args.loopback > 1):
        args.scaling = "strong"                   # configs[3] is ONE 512^3 grid cut into --gpus slabs, whatever N

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.loopback:
        # started directly (python bench.py --gpus N): become the launcher of N rank processes.  Nothing has touched the GPU
        # yet in this process; the ranks are CHILD processes (never an exec of this one) and their exit code is ours.
        import subprocess
        port = os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 2000))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # The first real N > 1 run is the driver's: it must not be able to hang.  The ranks run in their own process group under a
        # wall-clock budget; past it the whole group is killed and this launcher (which never touched the GPU) exits non-zero.
        # The ranks carry their own watchdog (below) with tighter, per-phase budgets, so this is the second line.
        budget = float(os.environ.get("FLUIDX_BENCH_TIMEOUT_S", "1500"))
        import signal
        proc = subprocess.Popen(cmd, env=env, start_new_session=True)
        try:
            rc_ = proc.wait(timeout=budget)
        except subprocess.TimeoutExpired:
            sys.stderr.write("bench.py: the %d rank processes did not finish within %.0f s (FLUIDX_BENCH_TIMEOUT_S): killing them\n" % (args.gpus, budget))
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            proc.wait()
            rc_ = 124
        raise SystemExit(rc_)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                         % (args.gpus, world, args.gpus, args.gpus))
    N = world
    G = args.grid
    loop = args.loopback if (world == 1 and args.loopback > 1 and not args.dry_run) else 0
    if loop:
        N = loop                                  # logical ranks; still one process, one GPU, no torch.distributed

    # ---- watchdog (N > 1): a rank stuck in a collective or a stream wait cannot report anything; a thread can.  Every phase arms a
    # deadline; past it the thread says which phase, on which rank, and ends the process (exit code 3) -- torch.distributed.run then
    # tears the other ranks down and the launcher exits non-zero.  FLUIDX_BENCH_WATCHDOG_S scales the budgets (seconds per phase).
    class Watchdog:
        def __init__(self, on):
            self.deadline, self.label, self.on = None, "", on
            self.base = float(os.environ.get("FLUIDX_BENCH_WATCHDOG_S", "120"))
            if on:
                import threading
                threading.Thread(target=self._run, daemon=True).start()

        def arm(self, label, factor=1.0):
            self.label, self.deadline = label, time.monotonic() + self.base * factor

        def disarm(self):
            self.deadline = None

        def _run(self):
            while True:
                time.sleep(0.25)
                d = self.deadline
                if d is not None and time.monotonic() > d:
                    sys.stderr.write("bench.py watchdog: rank %d stuck in phase '%s' for more than its budget -- exiting with code 3\n" % (rank, self.label))
                    sys.stderr.flush()
                    os._exit(3)

    watch = Watchdog(N > 1 and not loop and not args.dry_run)

    dist = None
    if N > 1 and not loop:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("NCCL_DEBUG", "WARN")               # RCCL's own complaints, on stderr (stdout carries the one JSON line)
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
        watch.arm("process group init")
        if args.dry_run or args.shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=N)
            if args.shared_gpu:
                local_rank = 0
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=N, device_id=torch.device("cuda", local_rank))

    (GX, GY, GZ), halo_adv = workload_grid(G, N, args.scaling)
    z0, nz = slab_for_rank(GZ, rank, N)

    fluid = None
    members = []
    if not args.dry_run:
        import fluidx12_amd as fx
        for r in ([rank] if not loop else range(N)):
            z0r, nzr = slab_for_rank(GZ, r, N)
            f_ = fx.Fluid()
            ok = f_.Init(1920, 1080, (GX, GY, GZ), storage=args.storage, jacobi_iters=args.iters, jacobi_mode=args.mode,
                         advect_address=args.address, device=local_rank if (N > 1 and not loop) else (r if (loop and args.peer_devices) else -1),
                         slab=(z0r, nzr) if N > 1 else None, halo_advect=halo_adv,
                         halo_jacobi=pressure_round(GX, GZ // N, args.iters) if N > 1 else 0)
            if not ok:
                raise SystemExit("Fluid.Init failed (status %d): the HIP library needs a MI355X" % f_.last_status)
            members.append(f_)
        fluid = members[0]                        # loop-back: the first context drives the group
        if loop:
            fx.comm_init_local(members, peer=args.group == "peer")

    # ---- RCCL rendezvous: rank 0 creates the unique id, torch.distributed broadcasts the bytes ----
    if N > 1 and not loop:
        import torch
        if args.dry_run:
            uid = bytes(range(128)) if rank == 0 else None
        else:
            uid = fx.comm_unique_id() if rank == 0 else None
        box = [uid]
        dist.broadcast_object_list(box, src=0)
        uid = box[0]
        assert isinstance(uid, (bytes, bytearray)) and len(uid) >= 128
        if not args.dry_run:
            watch.arm("fx_comm_init_rank (ncclCommInitRank x 2)")
            fluid.comm_init_rank(uid, rank, N)
        watch.disarm()

    dt = 2.0 / GY                                 # FluidX12.cpp:266

    step_log = []                                 # every frame index the run stepped with, in order: what the parity replay repeats

    def one_step(k):
        step_log.append(k)
        if fluid is not None:
            fluid.UpdateFrame(dt, k % 3)
            fluid.Simulate(k % 3)

    def barrier_sync():
        if fluid is not None:
            fluid.Synchronize()
        if dist is not None:
            if not (args.dry_run or args.shared_gpu):
                import torch
                torch.cuda.synchronize()
            dist.barrier()

    def max_over_ranks(seconds):
        if dist is None:
            return seconds
        import torch
        t_ = torch.tensor([seconds], dtype=torch.float64)
        if not (args.dry_run or args.shared_gpu):
            t_ = t_.cuda()
        dist.all_reduce(t_, op=dist.ReduceOp.MAX)      # identical on every rank afterwards
        return float(t_.item())

    # ---- N > 1: which slab schedule (what travels behind what; results are bit-identical for all of them) -------------------
    # Link bandwidth, RCCL launch latency and cross-stream event latency decide this, none of which a 1-GPU box can
    # measure, so the bench measures it where it runs: every candidate gets 1 settling + 3 timed steps (untimed region).
    schedule = None
    if N > 1 and fluid is not None:
        from fluidx12_amd import capi

        def apply(ov, rnd):
            for m_ in members:
                m_.set_option(capi.OPT_OVERLAP, ov)
                m_.set_option(capi.OPT_JACOBI_ROUND, rnd)

        if os.environ.get("FLUIDX_BENCH_ADAPTIVE", "1") == "0":        # A/B knob: always exchange halo_advect planes
            for m_ in members:
                m_.set_option(capi.OPT_ADAPTIVE_HALO, 0)

        if args.schedule != "auto":
            ov, rnd = (int(v) for v in args.schedule.split(","))
            apply(ov, rnd)
            schedule = {"overlap": ov, "jacobi_round": rnd, "picked": "pinned by --schedule"}
        else:
            tried = []
            kk = 0
            K = pressure_round(GX, GZ // N, args.iters)
            for ov, rnd in [(o, K if r == 8 else r) for o, r in SCHEDULE_CANDIDATES]:
                watch.arm("schedule candidate overlap=%d round=%d" % (ov, rnd))
                if os.environ.get("FLUIDX_BENCH_FAULT") == "stall:%d" % rank and ov == 1:
                    watch.disarm()                                  # fault injection (tests): this rank never arrives; its PEERS must notice
                    time.sleep(1e6)
                apply(ov, rnd)
                one_step(kk); kk += 1
                barrier_sync()
                for m_ in members:
                    m_.timing_enable(True); m_.timing_read(reset=True)
                t_ = time.perf_counter()
                for _ in range(3):
                    one_step(kk); kk += 1
                barrier_sync()
                el = max_over_ranks(time.perf_counter() - t_)      # identical on every rank => identical pick
                tm = (members[len(members) // 2] if loop else fluid).timing_read(reset=True)
                for m_ in members:
                    m_.timing_enable(False)
                c_ = {"overlap": ov, "jacobi_round": rnd, "ms_per_step": el / 3 * 1e3,
                      # this rank's (loop-back: an inner rank's) exchanges over the three steps: group calls, bytes sent, time on their stream
                      "exchange_calls_per_step": tm.exchange_calls / 3.0, "sent_MB_per_step": tm.exchange_bytes / 3.0 / 1e6,
                      "exchange_ms_per_step": tm.exchange_ms / 3.0}
                tried.append(c_)
                if rank == 0:                                       # one line per candidate as it completes (stderr: stdout carries the result line)
                    sys.stderr.write("bench.py candidate: " + json.dumps(c_) + "\n"); sys.stderr.flush()
            watch.disarm()
            best = min(tried, key=lambda c: c["ms_per_step"])
            apply(best["overlap"], best["jacobi_round"])
            schedule = {"overlap": best["overlap"], "jacobi_round": best["jacobi_round"],
                        "picked": "fastest of the candidates timed before the warm-up", "candidates": tried}
            # link model from the two SERIAL schedules (nothing hidden: exchange time = calls x latency + bytes / bandwidth; two
            # equations, two unknowns) -- replaces the 50 GB/s and 20 us per call DESIGN.md section 7 had to assume
            ser = [c for c in tried if c["overlap"] == 0 and c["exchange_calls_per_step"] > 0]
            if len(ser) >= 2 and ser[0]["exchange_calls_per_step"] != ser[1]["exchange_calls_per_step"]:
                a_, b_ = ser[0], ser[1]
                det = a_["exchange_calls_per_step"] * b_["sent_MB_per_step"] - b_["exchange_calls_per_step"] * a_["sent_MB_per_step"]
                if abs(det) > 1e-9:
                    lat_ms = (a_["exchange_ms_per_step"] * b_["sent_MB_per_step"] - b_["exchange_ms_per_step"] * a_["sent_MB_per_step"]) / det
                    per_mb = (a_["exchange_calls_per_step"] * b_["exchange_ms_per_step"] - b_["exchange_calls_per_step"] * a_["exchange_ms_per_step"]) / det
                    schedule["link_model"] = {"group_call_latency_us": lat_ms * 1e3, "GBps_per_rank_both_faces": (1.0 / per_mb) if per_mb > 0 else None,
                                              "from": "exchange_ms = calls x latency + MB / bandwidth over the serial candidates (0, %d) and (0, %d); "
                                                      "loop-back and mock runs measure copies, not links" % (a_["jacobi_round"], b_["jacobi_round"])}

    # Device wake-up.  A GPU that has idled through library load and context creation runs its first ~40 ms of work 3-5 % slower than
    # later (tools/step_time_profile.py: 256^3 steps 5..29 at 0.78-0.79 ms, from step 35 on 0.745-0.76; a SECOND context started from the
    # zero state in the same process runs 0.742-0.759 from its first step -- the device, not the data).  `--steps 20 --warmup 5` is 20 ms
    # of work: all of it inside that ramp.  So the same step runs on a scratch context (same grid, zero state, single domain) for 80 ms
    # first; the contract's W warm-up steps and K timed steps follow on the fresh context, frames W .. W + K - 1 as before.  Reported
    # as `device_preheat`; --no-preheat gives the cold figure.
    preheat = None
    if fluid is not None and not loop and args.preheat and not args.no_preheat:
        watch.arm("device wake-up", 2.0)
        import fluidx12_amd as fx_
        scratch = fx_.Fluid()
        if scratch.Init(1920, 1080, (GX, GY, nz if N > 1 else GZ), storage=args.storage, jacobi_iters=args.iters, jacobi_mode=args.mode,
                        advect_address=args.address, device=local_rank if N > 1 else -1):
            t_p, n_p = time.perf_counter(), 0
            while time.perf_counter() - t_p < 0.080 and n_p < 400:
                for _ in range(4):
                    scratch.UpdateFrame(dt, n_p % 3)
                    scratch.Simulate(n_p % 3)
                    n_p += 1
                scratch.Synchronize()
            preheat = {"steps": n_p, "ms": (time.perf_counter() - t_p) * 1e3,
                       "what": "the same step on a scratch context of the same grid before the warm-up steps: a device that has idled runs its first "
                               "~40 ms of work 3-5 % slower (tools/step_time_profile.py); --no-preheat = the cold figure"}
            scratch.Release()
        watch.disarm()

    watch.arm("warm-up", 1.0 + args.warmup / 50.0)
    for k in range(args.warmup):
        one_step(k)
    if fluid is not None:
        for m_ in members:
            m_.timing_enable(True)
            m_.timing_read(reset=True)
    barrier_sync()
    # The library's HIP-event marks (two event records per phase, round and member: the stream drains at each) are instrumentation
    # the product runs without; they cost a multi-rank step about a tenth (profiles/archive/r02d_comm_priority.txt: 5.1 -> 5.6 ms in
    # loop-back N = 4) and a single-GPU step 0.980 -> 0.947 ms at 256^3, 0.2047 -> 0.1814 ms at 128^3 (ten event records, ~2.5 us
    # each).  So only every fourth step of the timed region carries them; stage times, launch averages (roofline) and exchanged
    # bytes are per marked step, `value` is over all steps.  FLUIDX_BENCH_MARK_EVERY=1: every step.
    mark_every = 4 if args.steps >= 8 else 1
    if os.environ.get("FLUIDX_BENCH_MARK_EVERY"):
        mark_every = max(1, int(os.environ["FLUIDX_BENCH_MARK_EVERY"]))
    watch.arm("timed steps", 1.0 + args.steps / 50.0)
    t0 = time.perf_counter()
    for k in range(args.steps):
        if mark_every > 1 and k % mark_every in (0, 2, 3):             # marked: steps 2, 6, 10 ... (not the first one behind the barrier)
            for m_ in members:
                m_.timing_enable(k % mark_every == 2)
        one_step(args.warmup + k)
    barrier_sync()
    elapsed = time.perf_counter() - t0

    elapsed = max_over_ranks(elapsed)
    watch.disarm()

    # ---- N > 1: certify what was timed.  Every rank takes a device-side digest of its OWNED planes of velocity, colour and pressure
    # (fx_field_digest: a sum over the elements of a mix of (stored bits, global position) -- independent of the decomposition); rank 0
    # then replays the very same step sequence (schedule candidates, warm-up, timed steps) as ONE domain on its own GPU and takes the
    # digests of each rank's planes there.  Equal digests = bit-identical fields.  A mismatch is named in the line and is exit code 4.
    parity, parity_rc = None, 0
    if N > 1:
        if args.dry_run:
            parity = {"result": "not checked (dry run: no GPU work)"}
        elif args.no_parity:
            parity = {"result": "not checked (--no-parity)"}
        else:
            from fluidx12_amd import capi as capi_p
            FIELDS = (("velocity", capi_p.FIELD_VELOCITY), ("colour", capi_p.FIELD_COLOR), ("pressure", capi_p.FIELD_PRESSURE))
            watch.arm("parity: digests of the owned planes")
            fault = os.environ.get("FLUIDX_BENCH_FAULT", "")
            if fault.startswith("corrupt:"):          # fault injection (tests): one value of one rank's pressure flips a bit behind the timed steps
                import numpy as np
                victim = int(fault.split(":")[1])
                for r_, m_ in enumerate(members if loop else [fluid]):
                    if (r_ if loop else rank) == victim:
                        a_ = m_.download(capi_p.FIELD_PRESSURE)
                        a_.view(np.uint32)[a_.shape[0] // 2, a_.shape[1] // 2, a_.shape[2] // 2] ^= np.uint32(1)
                        m_.upload(capi_p.FIELD_PRESSURE, a_)
            mine = [[m_.digest(f_) for _, f_ in FIELDS] for m_ in (members if loop else [fluid])]
            if dist is not None:
                box = [None] * N
                dist.all_gather_object(box, mine[0])
                got = box
            else:
                got = mine
            t_par = time.perf_counter()
            verdict = None
            if rank == 0:
                watch.arm("parity: single-domain replay of %d steps" % len(step_log), 2.0 + len(step_log) * N / 100.0)
                import fluidx12_amd as fx_p
                ref = fx_p.Fluid()
                ok_ = ref.Init(1920, 1080, (GX, GY, GZ), storage=args.storage, jacobi_iters=args.iters, jacobi_mode=args.mode,
                               advect_address=args.address, device=local_rank if not loop else -1)
                if not ok_:
                    verdict = "single-domain replay could not be created (status %d)" % ref.last_status
                else:
                    for k_ in step_log:
                        ref.UpdateFrame(dt, k_ % 3)
                        ref.Simulate(k_ % 3)
                    ref.Synchronize()
                    for r_ in range(N):
                        z0r, nzr = slab_for_rank(GZ, r_, N)
                        for (name_, f_), d_ in zip(FIELDS, got[r_]):
                            if verdict is None and ref.digest(f_, z0r, nzr) != d_:
                                verdict = "rank %d: %s of planes [%d, %d) differs from the single-domain replay" % (r_, name_, z0r, z0r + nzr)
                    ref.Release()
                    verdict = verdict or "bit-identical"
            watch.disarm()
            if dist is not None:
                # the other ranks wait on the HOST (a key in the rendezvous store): in a device-side barrier they would spin on the GPUs
                try:
                    store_ = dist.distributed_c10d._get_default_store()
                    if rank == 0:
                        store_.set("fluidx_parity", verdict)
                    else:
                        import datetime
                        store_.wait(["fluidx_parity"], datetime.timedelta(seconds=600.0))
                        verdict = store_.get("fluidx_parity").decode()
                except Exception as e_:
                    box = [verdict]
                    dist.broadcast_object_list(box, src=0)
                    verdict = box[0]
            parity = {"result": verdict, "fields": [n_ for n_, _ in FIELDS], "steps_replayed": len(step_log), "ranks": N,
                      "replay_s": time.perf_counter() - t_par if rank == 0 else None,
                      "how": "fx_field_digest of every rank's owned planes (128 bits over stored bits x global position) == the same planes of ONE "
                             "domain stepped through the same %d frames on rank 0's GPU" % len(step_log)}
            if verdict != "bit-identical":
                parity_rc = 4

    # ---- the same W + K steps once more on a second context (zero state), now that the device is awake: `warm_device`.  `value` above is
    # the contract's protocol on whatever state the device was in when the process reached it (a GPU that has idled through library load
    # and context creation runs its first ~40 ms of work 3-5 % slower: tools/step_time_profile.py); this is the other truth, beside it.
    warm = None
    if fluid is not None and N == 1 and not loop and not args.no_warm_leg and not args.dry_run:
        import fluidx12_amd as fx_w
        w_ = fx_w.Fluid()
        if w_.Init(1920, 1080, (GX, GY, GZ), storage=args.storage, jacobi_iters=args.iters, jacobi_mode=args.mode, advect_address=args.address):
            for k_ in range(args.warmup):
                w_.UpdateFrame(dt, k_ % 3); w_.Simulate(k_ % 3)
            w_.Synchronize()
            t_w = time.perf_counter()
            for k_ in range(args.steps):
                w_.UpdateFrame(dt, (args.warmup + k_) % 3); w_.Simulate((args.warmup + k_) % 3)
            w_.Synchronize()
            el_w = time.perf_counter() - t_w
            warm = {"value": float(GX) * GY * GZ * args.steps / el_w, "unit": "voxel-updates/s", "ms_per_step": el_w / max(args.steps, 1) * 1e3,
                    "what": "the same %d warm-up + %d timed steps on a second context of the same grid (zero state), run right behind the timed region: "
                            "the device is awake; no library marks in this leg" % (args.warmup, args.steps)}
            w_.Release()

    roof = None
    timing = None
    render = None
    developed = None
    frame_now = args.warmup + args.steps
    if fluid is not None:
        timing = (members[len(members) // 2] if loop else fluid).timing_read(reset=True)   # loop-back: an inner rank's stages
        if N > 1 and schedule is not None and timing.steps:
            # what this rank (loop-back: an inner rank) sent across its faces, per face and step, and how many planes its advection
            # exchange carried (FX_OPT_ADAPTIVE_HALO: the measured need; halo_advect planes per face otherwise)
            my_rank = len(members) // 2 if loop else rank
            faces = (1 if my_rank > 0 else 0) + (1 if my_rank < N - 1 else 0)
            schedule["halo_advect_allocated_planes"] = halo_adv if halo_adv else 6
            schedule["advect_planes_per_face_and_step"] = timing.advect_halo_planes / max(timing.steps, 1) / max(faces, 1)
            schedule["sent_MB_per_face_and_step"] = timing.exchange_bytes / max(timing.steps, 1) / max(faces, 1) / 1e6
            schedule["exchange_calls_per_step"] = timing.exchange_calls / max(timing.steps, 1)
            schedule["exchange_ms_per_step"] = timing.exchange_ms / max(timing.steps, 1)
            if timing.exchange_ms > 0:                      # bytes this rank sent / the time its exchanges spent on their stream (latency included)
                schedule["link_GBps_sent_over_exchange_time"] = timing.exchange_bytes / (timing.exchange_ms * 1e-3) / 1e9
            schedule["adaptive_halo"] = os.environ.get("FLUIDX_BENCH_ADAPTIVE", "1") != "0"
        if N == 1 and G > 1 and not args.no_render:
            # config 3's second half, reported beside (never inside) `value`: the cube-map-space ray march of the state the
            # timed steps left behind -- default camera at 1920x1080 (FluidX12.cpp:243-253), OPTIMIZED = light volume + view pass
            import fluidx12_amd as fx
            view, proj, eye = fx.default_camera(1920, 1080)
            sh_info = None
            if args.config == 5:
                # BASELINE configs[4]'s second half: the SH light-probe GI path (CSSHCubeMap / Sum / Normalize) from the
                # synthetic 256^2 x 6 radiance cube L(dir) = max(dir.y, 0) * (1, .9, .8) + 0.1, then OPTIMIZED rendering with hasSH = 1
                import numpy as np
                n = 256
                idx = np.arange(n, dtype=np.float32)
                px_, py_ = np.meshgrid(idx - n / 2 + 0.5, -(idx - n / 2 + 0.5))       # the D3D cube-face convention of CubeMap.hlsli:5-35
                pz_ = np.full_like(px_, n / 2)
                dirs = [(pz_, py_, -px_), (-pz_, py_, px_), (px_, pz_, -py_), (px_, -pz_, py_), (px_, py_, pz_), (-px_, py_, -pz_)]
                cube = np.empty((6, n, n, 3), np.float32)
                for f_, (x_, y_, z_) in enumerate(dirs):
                    dy = y_ / np.sqrt(x_ * x_ + y_ * y_ + z_ * z_)
                    cube[f_] = np.maximum(dy, 0)[..., None] * np.array([1.0, 0.9, 0.8], np.float32) + np.float32(0.1)
                probe = fx.LightProbe(fluid)
                assert probe.Init(cube)
                probe.TransformSH(); fluid.Synchronize()
                t_sh = time.perf_counter()
                for _ in range(5):
                    probe.TransformSH()
                fluid.Synchronize()
                sh_info = {"radiance_cube": "synthetic 256^2 x 6, L = max(dir.y, 0) * (1, .9, .8) + 0.1",
                           "transform_ms_incl_upload": (time.perf_counter() - t_sh) / 5 * 1e3,
                           "sh_band0_rgb": [float(v) for v in probe.GetSH()[0]]}
                fluid.SetSH(probe.GetSH())
            # SURVEY 8(d) pins config 3's render to the state after 32 warm-up + 100 timed steps: the plume is what the marches cost, so the
            # figures are only comparable at a fixed frame.  Steps still missing to frame 132 are run here, untimed (the driver's
            # `--steps 20 --warmup 5` line used to render a barely formed plume: 0.156 / 0.133 ms against 0.248 / 0.177 at frame 116).
            RENDER_FRAME = 132
            steps_done = args.warmup + args.steps
            # (at least one: the step that makes the rendered frame follows a render, so that its advection writes the render's alpha side volume
            # -- what a frame loop does; the default 32 + 100 steps therefore render frame 133)
            extra = max(1, RENDER_FRAME - steps_done)
            from fluidx12_amd import capi as capi_
            # A context that renders its frames has the advection of the NEXT step write the render's alpha side volume (one more 4-byte
            # store per voxel, fx_advect_lds.hip <ALPHA>; the timed steps above rendered nothing and did not pay for it).  The frame before
            # RENDER_FRAME is therefore rendered once, untimed, and the step that makes RENDER_FRAME carries the marks: its advection stage
            # is reported beside the render (`advect_ms_of_a_rendered_frame`) -- the renders timed below cost what a render behind its own
            # step costs, and what they no longer do is paid for there.
            adv_rendered = None
            for k in range(extra):
                last = k == extra - 1
                if last:
                    fluid.UpdateFrame(0.0, 0, view, proj, eye)
                    fluid.Render(0, fx.Fluid.OPTIMIZED)
                    fluid.Synchronize()
                    fluid.timing_enable(True)
                    fluid.timing_read(reset=True)
                one_step(steps_done + k)
                if last:
                    fluid.Synchronize()
                    adv_rendered = fluid.timing_read(reset=True).advect_ms
            fluid.Synchronize()
            fluid.timing_enable(True)                               # (the timed loop leaves the marks off behind its last marked step)
            fluid.UpdateFrame(0.0, 0, view, proj, eye)
            fluid.Render(0, fx.Fluid.OPTIMIZED)
            fluid.Synchronize()
            fluid.timing_read(reset=True)
            # one COUNTED render (every thread adds its sample counts with atomics: statistics, never timed)
            fluid.set_option(capi_.OPT_COUNT_SAMPLES, 1)
            fluid.Render(0, fx.Fluid.OPTIMIZED)
            fluid.Synchronize()
            tc_ = fluid.timing_read(reset=True)
            fluid.set_option(capi_.OPT_COUNT_SAMPLES, 0)
            nr = 5
            fluid.ClearRenderTarget()
            for _ in range(nr):
                fluid.Render(0, fx.Fluid.OPTIMIZED, to_target=True)     # + renderCube: cube map -> 1920x1080 RGBA8 target
            fluid.Synchronize()
            tr_ = fluid.timing_read(reset=True)
            # the same passes by the PLAIN kernels (FX_OPT_RENDER_ACCEL 0: every sample gathers its taps, one lane per ray / voxel -- the
            # reference's own shape): the yardstick the default path is bit-identical to
            fluid.set_option(capi_.OPT_RENDER_ACCEL, 0)
            fluid.Render(0, fx.Fluid.OPTIMIZED)
            fluid.Synchronize()
            fluid.timing_read(reset=True)
            for _ in range(2):
                fluid.Render(0, fx.Fluid.OPTIMIZED)
            fluid.Synchronize()
            tp_ = fluid.timing_read(reset=True)
            fluid.set_option(capi_.OPT_RENDER_ACCEL, 1)
            # the paper's comparison (row f-2): the direct screen-space march of every pixel with the same light volume
            fluid.Render(0, fx.Fluid.SEPARATE_LIGHT_PASS)
            fluid.Synchronize()
            fluid.timing_read(reset=True)
            for _ in range(nr):
                fluid.Render(0, fx.Fluid.SEPARATE_LIGHT_PASS)
            fluid.Synchronize()
            td_ = fluid.timing_read(reset=True)
            fi = fluid.frame_info()
            rays = bin(fi.visibility_mask).count("1") * fi.cube_size ** 2
            cells_ = float(G) ** 3
            Cb_ = 16 if args.storage == "fp32" else 8
            light_s, view_s = tr_.light_ms / nr * 1e-3, tr_.view_ms / nr * 1e-3
            # compulsory traffic: the colour volume read once per pass, the light map written once by the light
            # pass and read once by the view pass, the cube map written once -- a lower bound the marches (cache / gather bound) are
            # measured against, not an HBM-fraction target
            light_bytes = (Cb_ + 4) * cells_
            view_bytes = (Cb_ + 4) * cells_ + 4.0 * rays
            rp = render_pmc(G, args.storage, args.config == 5)
            render = {"frame": steps_done + extra, "untimed_steps_to_frame": extra,
                      "mode": "OPTIMIZED (CSRayMarchL + CSRayMarchV) + renderCube (PSRayCastCube, raster-free)", "viewport": [1920, 1080], "cube_lod": fi.cube_lod,
                      "cube_size": fi.cube_size, "ray_samples": fi.ray_samples, "light_samples": 64, "rays": rays,
                      "light_pass_ms": tr_.light_ms / nr, "view_pass_ms": tr_.view_ms / nr,
                      "advect_ms_of_a_rendered_frame": adv_rendered,
                      "plain_kernels": {"light_pass_ms": tp_.light_ms / 2, "view_pass_ms": tp_.view_ms / 2,
                                        "note": "FX_OPT_RENDER_ACCEL 0: same pictures, bit for bit (tests/test_gpu_render.py::test_empty_space_skipping_changes_no_bit)"},
                      "cube_resolve_ms": tr_.resolve_ms / nr,
                      "direct_march_ms": td_.view_ms / nr, "direct_rays": 1920 * 1080,
                      "rays_per_s": rays / (tr_.view_ms / nr * 1e-3) if tr_.view_ms > 0 else None,
                      "light_voxels_per_s": float(G) ** 3 / (tr_.light_ms / nr * 1e-3) if tr_.light_ms > 0 else None,
                      # from ONE counted render of the same state (FX_OPT_COUNT_SAMPLES): trilinear colour fetches of the view rays, density
                      # fetches of the light pass (one per voxel + the light rays of the voxels with density >= 0.01), light-map fetches
                      "view_samples_taken": int(tc_.view_samples), "light_samples_taken": int(tc_.light_samples), "lightmap_fetches_taken": int(tc_.lightmap_fetches),
                      "view_samples_per_s": tc_.view_samples / view_s if view_s > 0 else None,
                      "light_samples_per_s": tc_.light_samples / light_s if light_s > 0 else None,
                      "samples_per_s": (tc_.view_samples + tc_.light_samples + tc_.lightmap_fetches) / (view_s + light_s) if view_s + light_s > 0 else None,
                      "mean_samples_per_view_ray": tc_.view_samples / max(rays, 1),
                      # what bounds the passes ("render"): the light pass = an HBM-bound build (every alpha read out of the
                      # colour texels once) + instruction issue of the ray kernels; the view pass = instruction issue (eight lanes per ray: the
                      # dependent chain of a ray is no longer the launch's duration).  The compulsory bytes are the floor both are held against.
                      "bound": {"kind": "light pass: VALU issue of the ray marches + a build pass over the alpha side volume (4 B per voxel; the advection of a rendered frame wrote it); view pass: VALU issue "
                                        "(k_view_slots); compulsory traffic is the lower bound, not an HBM-fraction target",
                                "light_pass": {"compulsory_bytes": light_bytes, "GBps": light_bytes / light_s / 1e9 if light_s > 0 else None,
                                               "frac_of_hbm_peak": light_bytes / light_s / 1e9 / HBM_PEAK_GBS if light_s > 0 else None},
                                "view_pass": {"compulsory_bytes": view_bytes, "GBps": view_bytes / view_s / 1e9 if view_s > 0 else None,
                                              "frac_of_hbm_peak": view_bytes / view_s / 1e9 / HBM_PEAK_GBS if view_s > 0 else None},
                                "cache": rp}}
            if sh_info is not None:
                render["mode"] += ", hasSH = 1 (light probe)"
                render["sh_light_probe"] = sh_info
            frame_now = steps_done + extra
        fluid.timing_enable(False)
        if N == 1 and G > 1 and not args.no_developed:
            # `value` is measured on the plume the warm-up left (the driver's line: steps 6-25, a young plume).  Two stages cost more on a
            # developed one: the advection (more waves trace beyond the staged window) and, in the reference's configuration, the pressure
            # solve (more cells keep relaxing).  So the same steps are timed once more past SURVEY 8(d)'s frame 132, beside `value`.
            # The launches right behind the render leg (its passes are separated by host synchronisations: the device idles between them)
            # run ~10 % slower whatever the plume -- round 6: Jacobi 0.46 against 0.41 ms per step for the 20 steps behind it, 0.406 forty
            # steps later; rounds 4-5 reported that transient as the developed plume's cost ("8-12 % slower") where the plume's own is
            # under 1 %.  So DEV_SETTLE steps run on a SCRATCH context first (same grid, zero state: the simulation under test is not
            # advanced, the frames stay SURVEY's 133-152).
            DEV_FRAME, DEV_STEPS, DEV_SETTLE = 132, 20, int(os.environ.get("FLUIDX_BENCH_DEV_SETTLE", "40"))
            for k in range(max(0, DEV_FRAME - frame_now)):
                one_step(frame_now + k)
            frame_now = max(frame_now, DEV_FRAME)
            if DEV_SETTLE:
                import fluidx12_amd as fx_s
                scr = fx_s.Fluid()
                if scr.Init(1920, 1080, (GX, GY, GZ), storage=args.storage, jacobi_iters=args.iters, jacobi_mode=args.mode, advect_address=args.address):
                    for k in range(DEV_SETTLE):
                        scr.UpdateFrame(dt, k % 3); scr.Simulate(k % 3)
                    scr.Synchronize()
                    scr.Release()
            fluid.Synchronize()
            fluid.timing_read(reset=True)
            t_dev = time.perf_counter()
            for k in range(DEV_STEPS):
                fluid.timing_enable(k % 4 == 2)                         # marks on every fourth step, as in the timed region
                one_step(frame_now + k)
            fluid.Synchronize()
            dev_s = (time.perf_counter() - t_dev) / DEV_STEPS
            t_dev_ = fluid.timing_read(reset=True)
            fluid.timing_enable(False)
            developed = {"frames": [frame_now + 1, frame_now + DEV_STEPS], "ms_per_step": dev_s * 1e3, "value": float(GX) * GY * nz / dev_s,
                         "unit": "voxel-updates/s", "settling_steps_behind_the_render_leg": DEV_SETTLE,
                         "stage_ms_per_step": {k_: getattr(t_dev_, k_ + "_ms") / max(t_dev_.steps, 1) for k_ in ("advect", "divergence", "jacobi", "project")},
                         "marked_steps": int(t_dev_.steps)}
            if t_dev_.freeze_solves:
                developed["sweeps_executed_per_solve"] = t_dev_.freeze_sweeps / t_dev_.freeze_solves
                developed["masked_strip_launches_per_solve"] = t_dev_.freeze_strip_launches / t_dev_.freeze_solves
            frame_now += DEV_STEPS
        if timing.jacobi_launches and args.mode == "faithful" and timing.freeze_solves:
            # The reference's own solve on the sparse solver (fx_jacobi_freeze.hip): one dense sweep (k_freeze_dense: the launch the
            # library books as "main") + tile launches over the cells that still relax.  roofline = the dense sweep, the one streaming
            # kernel of the phase with a fixed byte count: SURVEY 8(d)'s 12 B per cell-sweep (read p, read b, write p') x the cells of
            # ONE sweep.  By design it moves more (16.25 B: level 1 goes to two buffers, + two quad-nibble masks): `design_bytes`.
            cells = float(GX) * GY * nz
            steps_m = max(timing.steps, 1)
            dense_s = timing.jacobi_main_ms * 1e-3 / max(timing.jacobi_main_launches, 1)
            tile_l = int(timing.jacobi_launches - timing.jacobi_main_launches)
            tile_ms = timing.jacobi_ms - timing.jacobi_main_ms
            sweeps = timing.freeze_sweeps / timing.freeze_solves           # what the reference's loop executes per solve (<= iters)
            algo = JACOBI_BYTES_PER_CELL_SWEEP * cells
            design = 16.25 * cells
            # whole steps leave the divergence to this launch (no divergence launch was timed): SURVEY 8(d)'s V + S for it on top of the
            # sweep's 12 B -- the kernel reads the advected velocity instead of b and writes b for the tile launches
            fused_div = timing.divergence_ms == 0
            if fused_div:
                Vb = 12.0 if args.storage == "fp32" else 6.0
                algo += (Vb + 4.0) * cells
                design += (Vb + 4.0 - 4.0) * cells                     # + velocity, + b written, - b read
            tr = pmc_traffic("k_freeze_dense", G, args.iters, args.storage, "faithful") if N == 1 else None
            use_tr = tr is not None and not tr[2]
            roof = {"schema": 2,      # 2 (round 3 on): achieved / frac = COMPULSORY bytes of the launch (12 B x cells, whatever it fuses); the per-sweep figure of schema 1 is *_algorithmic
                    "bound": "hbm",
                    "kernel": "k_freeze_dense (%ssweep 1 of the reference's <= %d-sweep solve for every cell; sweeps 2.. run in %d tile launches of "
                              "k_freeze_tiles over the 32x8x8 tiles that still hold a relaxing cell: `sparse_solver`)" % ("the divergence + " if fused_div else "", args.iters, tile_l // steps_m),
                    "fused_divergence": fused_div,
                    "achieved": algo / dense_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algo / dense_s / 1e9 / HBM_PEAK_GBS,
                    "traffic": tr[0] if use_tr else None, "traffic_source": tr[1] if tr else None, "stale": bool(tr and tr[2]),
                    "algorithmic_bytes_per_launch": algo, "design_bytes_per_launch": design,
                    "frac_design_bytes": design / dense_s / 1e9 / HBM_PEAK_GBS,
                    "frac_traffic": (tr[0] / dense_s / 1e9 / HBM_PEAK_GBS) if use_tr else None,
                    "limiter": limiter_note("k_freeze_dense"),
                    "avg_launch_us": dense_s * 1e6, "launches": int(timing.jacobi_main_launches), "sweeps_per_launch": 1.0,
                    "sparse_solver": {
                        "sweeps_executed_per_solve": sweeps, "sweep_cap": args.iters, "solves": int(timing.freeze_solves),
                        "masked_strip_launches_per_solve": timing.freeze_strip_launches / timing.freeze_solves,   # which launch sequence the solves took (0 = tile launches only)
                        "tile_launches_per_step": tile_l / steps_m, "tile_launches_ms_per_step": tile_ms / steps_m,
                        "jacobi_phase_ms_per_step": timing.jacobi_ms / steps_m,
                        # the rate a dense replay of the executed sweeps would need to finish in the same time
                        "dense_equivalent_cell_updates_per_s": cells * sweeps / (timing.jacobi_ms / steps_m * 1e-3),
                        "dense_equivalent_GBps": JACOBI_BYTES_PER_CELL_SWEEP * cells * sweeps / (timing.jacobi_ms / steps_m * 1e-3) / 1e9,
                        "note": "bytes = 12 B x cells x the sweeps the reference's loop executes; above the HBM peak because most cells "
                                "leave the loop after the first sweep and are never touched again"}}
        elif timing.jacobi_launches:
            cells = float(GX) * GY * nz                                # cells this rank sweeps
            # the dominant kernel = the launches with the most sweeps each (a 40-sweep step is 12 launches of three + 2 of two);
            # the library books them separately, so the figure below is ONE kernel's average launch, as rocprofv3 reports it
            main_l = timing.jacobi_main_launches or timing.jacobi_launches
            main_ms = timing.jacobi_main_ms if timing.jacobi_main_launches else timing.jacobi_ms
            main_sw = timing.jacobi_main_sweeps if timing.jacobi_main_launches else timing.jacobi_sweeps
            avg_launch_s = main_ms * 1e-3 / main_l
            sweeps_per_launch = main_sw / main_l
            names = {1: ["k_jacobi_generic"] if (args.mode == "faithful" or GX % 4 or GZ == 1) else ["k_jacobi_v4"],
                     2: (["k_jacobi_strip2h"] if GX == 512 else ["k_jacobi_block2"] if GX == 128 else ["k_jacobi_blockg"] if GX not in (64, 256) else []) + ["k_jacobi_strip2u", "k_jacobi_strip"],
                     4: ["k_jacobi_strip4x"] if GX == 512 else ["k_jacobi_strip4t"] if GX != 256 else (["k_jacobi_strip4o"] if os.environ.get("FLUIDX_STRIP4_OCTET", "1") != "0" else []) + ["k_jacobi_strip4q"],
                     3: (["k_jacobi_strip3h"] if GX == 512 else ["k_jacobi_strip3c"] if (GX == 256 and GY % 8 == 0 and os.environ.get("FLUIDX_STRIP3_COOP", "1") != "0") else []) + ["k_jacobi_strip3", "k_jacobi_strip"]}
            cands = names.get(int(round(sweeps_per_launch)), ["k_jacobi_strip"]) if abs(sweeps_per_launch - round(sweeps_per_launch)) < 1e-9 else ["k_jacobi_strip"]
            tr = None
            if N == 1:
                for cand in cands:
                    tr = tr or pmc_traffic(cand, G, args.iters, args.storage)
            use_tr = tr is not None and not tr[2]
            tail_l = int(timing.jacobi_launches - main_l)
            # What bounds a T-sweep launch: it must read p and b once and write p' once whatever T is -- 12 B per cell (SURVEY 8(d)'s
            # per-sweep figure x the cells of ONE sweep).  `achieved` / `frac` are that: a fraction of the HBM peak that cannot exceed 1.
            # SURVEY 8(d)'s figure counted per fused sweep (12 B x cells x T: what T single-sweep launches would move) is kept beside it
            # as `achieved_algorithmic` / `frac_algorithmic`; it exceeds the peak by up to T x -- the temporal blocking, not a bound.
            compulsory = JACOBI_BYTES_PER_CELL_SWEEP * cells
            achieved = compulsory / avg_launch_s / 1e9
            roof = {"schema": 2,      # 2 (round 3 on): achieved / frac = COMPULSORY bytes of the launch (12 B x cells, whatever it fuses); the per-sweep figure of schema 1 is *_algorithmic
                    "bound": "hbm",
                    "kernel": "%s (one lock-step Jacobi sweep per launch)" % cands[0] if sweeps_per_launch == 1 else
                              "%s (%g lock-step Jacobi sweeps per launch, register/LDS-resident temporal blocking: p and b are read once "
                              "and p' written once per launch = the bytes `achieved` counts)" % (cands[0], sweeps_per_launch),
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS,
                    "traffic": tr[0] if use_tr else None,
                    "traffic_source": tr[1] if tr else None,
                    "stale": bool(tr and tr[2]),                     # the committed counter summary was taken on another version of this kernel
                    "compulsory_bytes_per_launch": compulsory,
                    "frac_compulsory": achieved / HBM_PEAK_GBS,
                    "frac_traffic": (tr[0] / avg_launch_s / 1e9 / HBM_PEAK_GBS) if use_tr else None,
                    "limiter": limiter_note(cands[0]),
                    "algorithmic_bytes_per_launch": JACOBI_BYTES_PER_CELL_SWEEP * cells * sweeps_per_launch,
                    "achieved_algorithmic": JACOBI_BYTES_PER_CELL_SWEEP * cells * sweeps_per_launch / avg_launch_s / 1e9,
                    "frac_algorithmic": JACOBI_BYTES_PER_CELL_SWEEP * cells * sweeps_per_launch / avg_launch_s / 1e9 / HBM_PEAK_GBS,
                    "avg_launch_us": avg_launch_s * 1e6, "launches": int(main_l),
                    "sweeps_per_launch": sweeps_per_launch,
                    "other_jacobi_launches": None if not tail_l else {
                        "launches": tail_l, "sweeps_per_launch": (timing.jacobi_sweeps - main_sw) / tail_l,
                        "avg_launch_us": (timing.jacobi_ms - main_ms) * 1e3 / tail_l},
                    "cell_updates_per_s": cells * timing.jacobi_sweeps / (timing.jacobi_ms * 1e-3)}

    # ---- N > 1 over RCCL: the same steps once more through the in-process peer transport (one process owning the N devices, halo planes
    # by hipMemcpyPeerAsync, no RCCL), as a CHILD process of rank 0 with its own deadline -- whatever happens to it, the RCCL line
    # stands.  The ranks give their memory back first and wait for rank 0 at the final barrier.
    peer_leg = None
    if N > 1 and not loop and dist is not None and not (args.dry_run or args.shared_gpu or args.no_peer_leg):
        for m_ in members:
            m_.Release()
        members = []
        dist.barrier()
        if rank == 0:
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--loopback", str(N), "--group", "peer", "--peer-devices", "--grid", str(args.grid),
                   "--iters", str(args.iters), "--storage", args.storage, "--mode", args.mode, "--address", args.address, "--scaling", args.scaling,
                   "--steps", str(args.steps), "--warmup", str(args.warmup), "--no-render", "--no-developed", "--no-cpu-baseline"]
            if args.config:
                cmd += ["--config", str(args.config)]
            if schedule is not None and "overlap" in schedule:
                cmd += ["--schedule", "%d,%d" % (schedule["overlap"], schedule["jacobi_round"])]
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE",
                                                                     "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
            try:
                r_ = subprocess.run(cmd, capture_output=True, text=True, timeout=float(os.environ.get("FLUIDX_BENCH_PEER_TIMEOUT_S", "300")), env=env)
                line = [ln for ln in r_.stdout.splitlines() if ln.startswith("{")]
                if r_.returncode == 0 and line:
                    d_ = json.loads(line[-1])
                    peer_leg = {"value": d_["value"], "unit": d_["unit"], "ms_per_step": d_["ms_per_step"], "devices_driven_by_one_process": N,
                                "ranks_seen": 1, "transport": "fx_comm_init_peer: hipMemcpyPeerAsync between the slabs' devices, one process, no RCCL",
                                "schedule": d_["config"].get("schedule"), "stage_ms_per_step": d_.get("stage_ms_per_step")}
                else:
                    peer_leg = {"error": "exit code %d" % r_.returncode, "stderr_tail": r_.stderr[-400:]}
            except subprocess.TimeoutExpired:
                peer_leg = {"error": "timeout"}
            except Exception as e_:                          # never lets the RCCL line down
                peer_leg = {"error": repr(e_)}
            if isinstance(peer_leg, dict):
                peer_leg["note"] = ("timed with the rank processes still attached to their devices but parked on the HOST (a key in the rendezvous "
                                    "store, not a device-side barrier); until a node run has been inspected this is a functional figure, not a measurement")
        # The other ranks wait for the child on the host: in the final barrier of the NCCL group they would spin in a device kernel on the
        # very GPUs the child is timing.  The rendezvous store carries the word.
        try:
            store_ = dist.distributed_c10d._get_default_store()
            if rank == 0:
                store_.set("fluidx_peer_leg_done", "1")
            else:
                import datetime
                store_.wait(["fluidx_peer_leg_done"], datetime.timedelta(seconds=float(os.environ.get("FLUIDX_BENCH_PEER_TIMEOUT_S", "300")) + 60.0))
        except Exception as e_:                              # (no store / time-out: fall through to the barrier, as before)
            print("bench.py: host-side wait for the peer leg failed on rank %d: %r" % (rank, e_), file=sys.stderr, flush=True)

    if rank == 0:
        voxels = float(GX) * GY * GZ * args.steps
        out = {
            "metric": "voxel-updates/sec (advect+40 Jacobi) at 256^3; achieved HBM GB/s vs peak" if args.mode == "fixed" else
                      "voxel-updates/sec (advect + the reference's <= %d-sweep early-out Jacobi) at %d^3; achieved HBM GB/s vs peak" % (args.iters, G),
            "value": voxels / elapsed if not args.dry_run else 0.0,
            "unit": "voxel-updates/s",
            "n_gpus": N if (not loop or args.peer_devices) else 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
            "higher_is_better": True,
            "scaling": ("weak" if args.scaling == "weak256" else args.scaling) if N > 1 else "weak",
            "scaling_table": args.scaling if N > 1 else None,
            "vs_baseline": None,
            "dtype": "f32" if args.storage == "fp32" else "f32 arithmetic / f16 field storage",
            "value_cold": (voxels / elapsed if not args.dry_run else 0.0) if preheat is None else None,   # = `value` unless --preheat was asked for
            "warm_device": warm,
            "device_preheat": preheat,
            "data": "synthetic" if not loop else ("synthetic; IN-PROCESS PEER GROUP: one process drives %d GPUs, a slab each" % loop) if args.peer_devices else
                    "synthetic; LOOP-BACK: %d slab ranks share ONE GPU (functional check of the multi-rank path, not a scaling measurement)" % loop,
            "config": {"workload": "%dx%dx%d 3D smoke (%.1f M voxels per GPU), %s, %s fields, %s sampler, "
                                   "advect+divergence+Jacobi+project per step; %s"
                                   % (GX, GY, GZ, GX * GY * GZ / N / 1e6,
                                      "%d Jacobi sweeps" % args.iters if args.mode == "fixed" else "the reference's pressure loop (<= %d sweeps, per-cell early-out at |dx| < 1e-3)" % args.iters,
                                      args.storage, args.address.upper(),
                                      baseline_config_label(GX, GY, GZ, args.iters, args.storage, N) if args.mode == "fixed" else
                                      "the REFERENCE's own configuration (CSProject3D.hlsl:13, CSPoisson.hlsli:8-26%s), not a BASELINE.json config (those fix the sweep count)"
                                      % (", RGBA16F fields Fluid.cpp:207-213" if args.storage == "fp16" else "")),
                       "grid": [GX, GY, GZ], "jacobi_iters": args.iters, "jacobi_mode": args.mode, "storage": args.storage, "address": args.address,
                       "parallelism": "single GPU" if N == 1 else ("z-slab x%d (%d planes per rank), " % (N, GZ // N)) + (("peer copies between the devices of one process (fx_comm_init_peer)" if args.peer_devices else "loop-back copies on one GPU, %s group" % args.group) if loop
                                                                                                                             else "RCCL send/recv halo exchange, %d rank processes" % N),
                       "schedule": schedule,
                       "bytes_per_voxel_step": step_bytes_per_voxel(args.iters, args.storage)},
        }
        if args.dry_run:
            out["dry_run"] = True
        if args.shared_gpu and N > 1:
            out["shared_gpu"] = True
            out["data"] = "synthetic; SHARED GPU: %d rank processes on ONE device (functional check of the one-process-per-GPU path, not a scaling measurement)" % N
        if timing is not None:
            out["timing_marks"] = {"every": mark_every, "marked_steps": int(timing.steps)}      # stage / launch figures are per marked step
            out["stage_ms_per_step"] = {k: getattr(timing, k + "_ms") / max(timing.steps, 1)
                                        for k in ("advect", "divergence", "jacobi", "project", "exchange")}
            eff_iters = args.iters if args.mode == "fixed" else (timing.freeze_sweeps / timing.freeze_solves if timing.freeze_solves else args.iters)
            sb = step_bytes_per_voxel(eff_iters, args.storage) * float(GX) * GY * GZ
            out["step_algorithmic_GBps"] = sb / (elapsed / args.steps) / 1e9     # 5V + 2C + 2S + 12 N per voxel (N = sweeps executed); > peak under temporal blocking
            if args.mode == "fixed" and roof is not None:
                # the bytes a step cannot avoid with the launch shapes it used: every Jacobi LAUNCH reads p, b and writes p' once
                V_, C_ = (12, 16) if args.storage == "fp32" else (6, 8)
                launches_per_step = timing.jacobi_launches / max(timing.steps, 1)
                sc = (5 * V_ + 2 * C_ + 2 * 4 + 12 * launches_per_step) * float(GX) * GY * GZ
                out["step_compulsory"] = {"bytes_per_step": sc, "GBps": sc / (elapsed / args.steps) / 1e9,
                                          "frac_of_peak": sc / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                                          "jacobi_launches_per_step": launches_per_step}
            st = step_traffic(G, args.iters, args.storage, args.mode) if N == 1 else None
            if st:
                # the whole step against the fabric: measured bytes of all its launches (PMC, committed summary) / measured time
                out["step_fabric_traffic"] = {"bytes_per_step": st[0] if not st[2] else None, "GBps": st[0] / (elapsed / args.steps) / 1e9 if not st[2] else None,
                                              "frac_of_peak": st[0] / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS if not st[2] else None, "source": st[1],
                                              "stale_kernels": st[2]}
        if roof is not None:
            out["roofline"] = roof
        if render is not None:
            out["render"] = render
        if developed is not None:
            out["developed_plume"] = developed
        if peer_leg is not None:
            out["peer_transport"] = peer_leg              # beside the RCCL figure in `value`, never instead of it
        if parity is not None:
            out["multi_rank_parity"] = parity["result"]
            out["multi_rank_parity_detail"] = parity
        if N > 1 and dist is not None and not args.dry_run:
            try:
                import torch
                out["rccl"] = {"version": ".".join(str(v_) for v_ in torch.cuda.nccl.version()) if not args.shared_gpu else "mock (FLUIDX_RCCL_LIB)",
                               "ranks_seen_by_torch_distributed": dist.get_world_size(), "backend": dist.get_backend()}
            except Exception as e_:
                out["rccl"] = {"error": repr(e_)}
        if N == 1 and not args.no_cpu_baseline and not args.dry_run:
            out["cpu_baseline"] = cpu_baseline(G, args.iters, args.cpu_budget, mode=int(args.mode == "faithful"), half=args.storage == "fp16",
                                               address=int(args.address == "mirror"))
        print(json.dumps(out), flush=True)

    for m_ in members:
        m_.Release()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if parity_rc:
        sys.stderr.write("bench.py: multi-rank parity FAILED: %s\n" % parity["result"])
        raise SystemExit(parity_rc)


if __name__ == "__main__":
    main()
