"""Can two RCCL ranks share ONE GPU on this box?  (decides whether world_size-2 RCCL tests can run on a 1-GPU box)"""
import os
import torch
import torch.distributed as dist
rank = int(os.environ['RANK'])
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
t = torch.full((4,), float(rank + 1), device='cuda', dtype=torch.float64)
dist.all_reduce(t)
torch.cuda.synchronize()
print('rank', rank, 'all_reduce ->', t.tolist(), flush=True)
dist.destroy_process_group()
