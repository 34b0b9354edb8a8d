"""One rank of a multi-PROCESS z-slab run on a single GPU (launched by tests/test_gpu_rccl_mock.py under
torch.distributed.run with FLUIDX_RCCL_LIB = the mock of tests/mock_rccl).  Every rank owns one slab context bound
through fx_comm_init_rank -- the RcclTransport code path the 8-GPU bench uses -- steps it under each slab schedule,
and the colour field gathered on rank 0 (fx_comm_gather_color) must equal the single-domain run bit for bit."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch.distributed as dist

import bench
import fluidx12_amd as fx
from fluidx12_amd import capi


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    X, Y, Z = (int(v) for v in sys.argv[1].split("x"))
    steps, iters, halo_j, storage = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    slabs = [bench.slab_for_rank(Z, r, world) for r in range(world)]
    dt = np.float32(2.0 / Y)
    failures = []
    for ov, rnd in ((0, halo_j), (1, halo_j), (2, halo_j), (3, halo_j), (2, max(1, halo_j // 2))):
        f = fx.Fluid()
        assert f.Init(640, 480, (X, Y, Z), storage=storage, jacobi_iters=iters, slab=slabs[rank], halo_advect=16,
                      halo_jacobi=halo_j, device=0), f.last_status
        uid = [fx.comm_unique_id() if rank == 0 else None]   # a fresh communicator per schedule (the previous one is released)
        dist.broadcast_object_list(uid, src=0)
        f.comm_init_rank(uid[0], rank, world)
        f.set_option(capi.OPT_OVERLAP, ov)
        f.set_option(capi.OPT_JACOBI_ROUND, rnd)
        for k in range(steps):
            f.UpdateFrame(dt, k % 3)
            f.Simulate(k % 3)
        full = None
        if rank == 0:
            full = fx.Fluid()
            assert full.Init(640, 480, (X, Y, Z), storage=storage, jacobi_iters=iters, device=0)
        f.gather_color(full, root=0, slabs=slabs)
        f.Synchronize()
        if rank == 0:
            got = full.download(fx.FIELD_COLOR)
            ref = fx.Fluid()
            assert ref.Init(640, 480, (X, Y, Z), storage=storage, jacobi_iters=iters, device=0)
            for k in range(steps):
                ref.UpdateFrame(dt, k % 3)
                ref.Simulate(k % 3)
            ref.Synchronize()
            want = ref.download(fx.FIELD_COLOR)
            if not (got.view(np.uint8) == want.view(np.uint8)).all() or not want.any():
                failures.append((ov, rnd, float(np.abs(got.astype(np.float64) - want).max())))
            ref.Release()
            full.Release()
        dist.barrier()                                      # nobody tears its communicator down while a peer still reads
        f.Release()
        dist.barrier()
    flag = [failures]
    dist.broadcast_object_list(flag, src=0)
    dist.destroy_process_group()
    if flag[0]:
        print("MISMATCH", flag[0], flush=True)
        sys.exit(1)
    if rank == 0:
        print("OK %d ranks, 5 schedules bit-identical" % world, flush=True)


if __name__ == "__main__":
    main()
