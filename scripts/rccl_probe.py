"""One-rank RCCL probe: the calls bench.py makes for N > 1 (group init with device_id, barrier, reduce of the float4
accumulator, all_reduce of float64 scalars) on a real device.  Run under torch.distributed.run with one process."""
import os
import torch
import torch.distributed as dist

local = int(os.environ.get('LOCAL_RANK', '0'))
torch.cuda.set_device(local)
dist.init_process_group('nccl', device_id=torch.device('cuda', local))
acc = torch.ones(1920 * 1080 * 4, dtype=torch.float32, device='cuda')
torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
dist.reduce(acc, dst=0, op=dist.ReduceOp.SUM)
t = torch.tensor([1.5], dtype=torch.float64, device='cuda'); dist.all_reduce(t, op=dist.ReduceOp.MAX)
r = torch.tensor([2.0, 3.0], dtype=torch.float64, device='cuda'); dist.all_reduce(r, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
print('rccl ok', float(acc.sum().item()), t.item(), r.tolist(), flush=True)
dist.barrier(); dist.destroy_process_group()
