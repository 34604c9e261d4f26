"""World-size-1 probe of the torch.distributed calls bench.py makes at N > 1 (RCCL init with device_id, barrier
with device_ids, MAX all-reduce, teardown) - the development box has one GPU."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=device)
dist.barrier(device_ids=[0])
torch.cuda.synchronize()
t = torch.tensor([1.5, 2.5], device=device, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("all_reduce ok", t.tolist())
dist.barrier(device_ids=[0])
dist.destroy_process_group()
print("nccl probe ok")
