"""Can two RCCL ranks share ONE GPU on this pool?  (decides whether the 2-rank RcclTransport can be exercised on a 1-GPU box)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 tools/rccl_dup_probe.py"""
import os
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
x = torch.full((1024,), float(rank + 1), device="cuda:0")
y = torch.zeros_like(x)
peer = 1 - rank
ops = [dist.P2POp(dist.isend, x, peer), dist.P2POp(dist.irecv, y, peer)]
for w in dist.batch_isend_irecv(ops):
    w.wait()
torch.cuda.synchronize()
print("rank", rank, "received", y[0].item(), flush=True)
dist.destroy_process_group()
