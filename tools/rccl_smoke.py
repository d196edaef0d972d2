"""One process, backend "nccl" (= RCCL), world size 1: the collectives the sharded mapping iteration and bench.py issue,
issued the way they are issued there, go through RCCL once on this box --

  * a float32 SUM all-reduce of a flat bucket of which the operands are in-place SLICES (backend_map.FlatReducer.sum_floats: the
    backward has written the gradients straight into the bucket);
  * an int32 MAX all-reduce and a uint8 MAX all-reduce (the byte-wise OR of the visibility flags), both STARTED
    asynchronously on a second communicator (dist.new_group: map_window's aux_group) before the float SUM is issued on the
    first and waited for after it;
  * the same through backend_map's own code (FlatReducer / _max_bytes) on a toy bucket;
  * a float64 MAX (bench.py's timing reduction) and a barrier.

Several ranks cannot share one GPU under RCCL; the multi-rank logic itself is covered by the gloo tests (world sizes 2-8).
    python tools/rccl_smoke.py"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
aux = dist.new_group()

N = 1 << 18
bucket = torch.empty(20 * N + 1, dtype=torch.float32, device=dev)
bucket.copy_(torch.arange(bucket.numel(), dtype=torch.float32, device=dev) % 1024)
grads = bucket[:14 * N]                       # in-place slices, as the backward leaves them
radii = torch.arange(N, dtype=torch.int32, device=dev) - 5
flags = (torch.arange(8 * N, device=dev) % 2).to(torch.uint8).view(8, N)
w1 = dist.all_reduce(radii, op=dist.ReduceOp.MAX, group=aux, async_op=True)
w2 = dist.all_reduce(flags[:6], op=dist.ReduceOp.MAX, group=aux, async_op=True)   # (a slice of the rows: the window's)
dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
w1.wait(); w2.wait()
t = torch.tensor([1.5], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert float(grads[12345]) == 12345 % 1024 and int(radii[0]) == -5 and float(t) == 1.5 and int(flags.sum()) == 4 * N

# ... and through the product's own reducer
import lvdgs  # noqa: E402,F401
from lvdgs.backend_map import FlatReducer, _max_bytes  # noqa: E402
red = FlatReducer()
pieces = red.plan_floats([3 * N, N, 1], dev)
for k, p in enumerate(pieces):
    p.fill_(float(k + 1))
out = red.sum_floats(pieces, [3 * N, N, 1], dev)
(ri,), work = red.max_ints([radii.clone()], dev, aux, async_op=True)
fw = _max_bytes(flags, aux, async_op=True)
for w in (work, fw):
    if w is not None:
        w.wait()
torch.cuda.synchronize()
assert [float(o[0]) for o in out] == [1.0, 2.0, 3.0] and int(ri[7]) == 2 and all(o.data_ptr() == p.data_ptr() for o, p in zip(out, pieces))
print("rccl ok: float32 SUM on in-place bucket slices, int32 MAX + uint8 MAX started asynchronously on a second communicator, "
      "FlatReducer / _max_bytes, float64 MAX, barrier")
dist.destroy_process_group()
