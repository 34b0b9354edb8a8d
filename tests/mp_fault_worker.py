"""One rank of a two-PROCESS slab run on a single GPU whose link fails (tests/test_gpu_rccl_mock.py; the product's RCCL calls bound to
tests/mock_rccl through FLUIDX_RCCL_LIB).  No torch: the unique id travels through a file.

    python tests/mp_fault_worker.py <rank> <id-file> <mode>

mode "async": the mock reports an asynchronous communicator error on rank 1 (FXMOCK_ASYNC_ERROR); "die": rank 1 ends abruptly after
three steps (os._exit: no release, no abort).  Either way EVERY surviving rank must come back with FX_E_COMM -- from fx_simulate or
fx_synchronize -- within seconds, stay failed on the next call, and release its context without hanging.  Prints one line and exits 0."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import fluidx12_amd as fx
from fluidx12_amd import capi


def main():
    rank, id_file, mode = int(sys.argv[1]), sys.argv[2], sys.argv[3]
    Z = 64
    f = fx.Fluid()
    assert f.Init(640, 480, (64, 64, Z), jacobi_iters=16, slab=(rank * Z // 2, Z // 2), halo_advect=16, halo_jacobi=8, device=0), f.last_status
    if rank == 0:
        uid = fx.comm_unique_id()
        with open(id_file + ".tmp", "wb") as fh:
            fh.write(uid)
        os.rename(id_file + ".tmp", id_file)
    else:
        t0 = time.monotonic()
        while not os.path.exists(id_file):
            assert time.monotonic() - t0 < 60
            time.sleep(0.01)
        uid = open(id_file, "rb").read()
    f.comm_init_rank(uid, rank, 2)
    dt = np.float32(2.0 / 64)
    t0 = time.monotonic()
    status, where, step = 0, "", -1
    try:
        for step in range(200):
            if mode == "die" and rank == 1 and step == 3:
                os._exit(0)                               # a rank that dies: nothing is released, nothing aborted
            f.UpdateFrame(dt, step % 3)
            where = "simulate"
            f.Simulate(step % 3)
            where = "synchronize"
            f.Synchronize()
    except capi.FluidxError as e:
        status = e.status
    took = time.monotonic() - t0
    again = 0
    try:                                                  # the communicator stays failed: the next call returns at once
        f.UpdateFrame(dt, 0)
        f.Simulate(0)
    except capi.FluidxError as e:
        again = e.status
    f.Release()
    print("RANK %d status %d in %s at step %d after %.1f s; again %d" % (rank, status, where, step, took, again), flush=True)


if __name__ == "__main__":
    main()
