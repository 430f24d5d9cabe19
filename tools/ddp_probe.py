"""RCCL all-reduce of gradient-buffer-sized tensors on the ranks torchrun started (world 1 on a one-GPU box: what the
exchange itself costs when nothing overlaps it).  python -m torch.distributed.run --nproc-per-node N tools/ddp_probe.py"""
import os, time
import torch
import torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dist.init_process_group("nccl")
for mb in (58, 116, 464):
    g = torch.ones(mb * 1024 * 1024 // 4, device="cuda")
    for _ in range(3):
        dist.all_reduce(g)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10):
        dist.all_reduce(g)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 10
    if dist.get_rank() == 0:
        print("all_reduce %4d MB: %.3f ms  (%.1f GB/s algorithmic)" % (mb, dt * 1e3, mb / 1024 / dt), flush=True)
dist.destroy_process_group()
