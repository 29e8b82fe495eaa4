"""RCCL plumbing check on one GPU (world size 1): process-group init with device_id, the tuple all-gather of shard.py on
uint8 / int32 device tensors, barrier, all_reduce(MAX) -- the calls bench.py makes for N > 1.
    python3 tools/rccl_smoke.py"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c6amd_loader import load_package  # noqa: E402

pkg = load_package()
from alphazero_connect6_amd import shard  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
stride = pkg.lib().c6_tuple_stride(19)
buf = torch.arange(4096 * stride, dtype=torch.int64, device="cuda").to(torch.uint8).reshape(4096, stride)
cnt = torch.tensor([17], dtype=torch.int32, device="cuda")
pooled = torch.empty((1,) + tuple(buf.shape), dtype=buf.dtype, device="cuda")
counts = torch.empty(1, dtype=torch.int32, device="cuda")
dist.all_gather_into_tensor(pooled.view(-1), buf.reshape(-1).contiguous())
dist.all_gather_into_tensor(counts, cnt.reshape(1))
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
assert torch.equal(pooled[0], buf) and int(counts[0]) == 17 and float(t) == 1.5
assert shard.compact(pooled, counts).shape == (17, stride)
dist.destroy_process_group()
print("rccl smoke ok: all_gather_into_tensor(uint8 %d B, int32), barrier, all_reduce(MAX)" % buf.numel())
