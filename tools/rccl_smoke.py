"""One process, backend "nccl" (= RCCL), world size 1: the collectives the sharded mapping iteration and bench.py issue --
a float32 SUM all-reduce and an int32 MAX all-reduce on flat buckets (backend_map.FlatReducer), a uint8 MAX (the byte-wise
OR of the visibility flags), a float64 MAX (the timing reduction) and a barrier -- go through RCCL once on this box.  (Several ranks cannot share one GPU under RCCL; the
multi-rank logic itself is covered by the gloo tests.)    python tools/rccl_smoke.py"""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
f = torch.arange(1 << 20, dtype=torch.float32, device=dev)
i = torch.arange(1 << 18, dtype=torch.int32, device=dev) - 5
t = torch.tensor([1.5], device=dev, dtype=torch.float64)
b = (torch.arange(1 << 18, device=dev) % 2).to(torch.uint8).view(8, -1)
dist.all_reduce(f, op=dist.ReduceOp.SUM)
dist.all_reduce(i, op=dist.ReduceOp.MAX)
dist.all_reduce(b, op=dist.ReduceOp.MAX)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert float(f[12345]) == 12345.0 and int(i[0]) == -5 and float(t) == 1.5 and int(b.sum()) == 1 << 17
print("rccl ok: float32 SUM, int32 MAX, uint8 MAX, float64 MAX, barrier")
dist.destroy_process_group()
