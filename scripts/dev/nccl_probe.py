"""GPU box probe: does backend="nccl" (RCCL) initialise at world 1, and do all_reduce / a side stream work against it?"""
import os
import torch
import torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
x = torch.arange(8, dtype=torch.float32, device="cuda")
dist.all_reduce(x)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    y = x * 2
    dist.all_reduce(y)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("nccl world 1 ok", x.tolist(), y.tolist(), "group2", dist.new_group() is not None)
dist.barrier()
dist.destroy_process_group()
