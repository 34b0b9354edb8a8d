"""Per-rank cost of the z-slab schedule, measured on ONE GPU through the loop-back transport.

All N slab contexts of a weak-scaling workload (bench.workload_grid) live in one process on one device and step
sequentially; wall time / N is what one rank spends on its own kernels plus the device-to-device halo copies --
i.e. the slab run WITHOUT link time.  Comparing with the single-domain 256^3 step gives the compute-side overhead of
the decomposition (halo-plane recompute, extra launches, thinner strip chunks); the xGMI time comes on top and is
estimated in DESIGN.md section 7 from the exchanged bytes.

  python tools/slab_loopback_bench.py [--gpus 2 4 8] [--steps 10]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
import fluidx12_amd as fx


def run(N, G, iters, steps, warmup, scaling, overlap=True, halo_jacobi=0):
    (X, Y, Z), halo = bench.workload_grid(G, N, scaling)
    fl = []
    for r in range(N):
        z0, nz = bench.slab_for_rank(Z, r, N)
        f = fx.Fluid()
        ok = f.Init(1920, 1080, (X, Y, Z), jacobi_iters=iters, slab=(z0, nz) if N > 1 else None, halo_advect=halo,
                    overlap=overlap, halo_jacobi=halo_jacobi)
        assert ok, f.last_status
        fl.append(f)
    if N > 1:
        fx.comm_init_local(fl)
    dt = np.float32(2.0 / Y)
    k = 0
    for _ in range(warmup):
        fl[0].UpdateFrame(dt, k % 3); fl[0].Simulate(k % 3); k += 1
    fl[0].Synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fl[0].UpdateFrame(dt, k % 3); fl[0].Simulate(k % 3); k += 1
    fl[0].Synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    for f in fl:
        f.Release()
    return (X, Y, Z), ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", default="weak")
    ap.add_argument("--halo-jacobi", type=int, default=0)
    a = ap.parse_args()
    base = None
    for N in a.gpus:
        for overlap in ((True,) if N == 1 else (True, False)):
            dims, ms = run(N, a.grid, a.iters, a.steps, a.warmup, a.scaling, overlap, a.halo_jacobi)
            per_rank = ms / N
            if N == 1:
                base = per_rank
            print("N=%d grid=%dx%dx%d %s all ranks on one GPU: %.3f ms/step  per rank %.3f ms%s" % (
                N, dims[0], dims[1], dims[2], "overlap " if overlap else "serial  ", ms, per_rank,
                "  (x%.3f of the single-domain step)" % (per_rank / base) if base else ""), flush=True)


if __name__ == "__main__":
    main()
