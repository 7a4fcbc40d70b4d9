"""RCCL smoke on one GPU: a world_size-1 "nccl" group running the exact collectives of the sharded
gate (uint8 all_gather_into_tensor carrying seams + threshold, barrier, float64 all_reduce)."""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from noisereduce_amd.sharded import exchange_seams_and_threshold
dev = torch.device("cuda", 0)
y = torch.randn(2, 100000, device=dev)
thr = torch.arange(513, dtype=torch.float64, device=dev)
left, right, t = exchange_seams_and_threshold(y, 3000, thr, 513, None, {})
torch.cuda.synchronize()
assert torch.equal(t, thr) and float(left.abs().max()) == 0.0 and float(right.abs().max()) == 0.0
x = torch.ones(4, device=dev, dtype=torch.float64); dist.all_reduce(x); dist.barrier()
s = torch.tensor([1.5], device=dev, dtype=torch.float64); dist.all_reduce(s, op=dist.ReduceOp.MAX)
print("nccl single-rank collectives ok", float(s))
dist.destroy_process_group()
