"""One-rank rehearsal of the collectives bench.py uses for N > 1 (process group "nccl" = RCCL, tensors on the GPU):
the strip gather, the MAX/MIN/SUM reductions and the barrier.  With one rank it proves that RCCL initialises on the
box and that every call accepts device tensors; the data path between GPUs needs a multi-GPU node.
usage: python tools/rccl_selfcheck.py"""
import os, sys
sys.path.insert(0, '.')
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
send = torch.arange(8 * 16 * 4, dtype=torch.uint8, device=dev).view(8, 16, 4)
recv = [torch.zeros_like(send)]
dist.gather(send, recv, dst=0)
assert torch.equal(recv[0], send)
x = torch.tensor([1.5, 2.5], dtype=torch.float64, device=dev)
dist.all_reduce(x, op=dist.ReduceOp.MAX); dist.all_reduce(x, op=dist.ReduceOp.MIN)
n = torch.tensor([7], dtype=torch.int64, device=dev); dist.all_reduce(n, op=dist.ReduceOp.SUM)
dist.barrier(); torch.cuda.synchronize()
assert x.tolist() == [1.5, 2.5] and int(n) == 7
dist.destroy_process_group()
print("rccl one-rank rehearsal ok")
