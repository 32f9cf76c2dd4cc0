#!/usr/bin/env python3
"""The collectives bench.py issues when N > 1 (barrier, all_reduce MAX of a float64 on the device, all_gather of the
packed record tensor), through RCCL with whatever world size the launcher provides -- a 1-GPU box can at least run it
with `python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/rccl_sanity.py`."""
import os, time
import torch
import torch.distributed as dist
rank, world, lr = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(lr)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", lr))
dev = torch.device("cuda", lr)
dist.barrier()
t = torch.tensor([1.5 + rank], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
rec = torch.full((52428800,), rank, dtype=torch.uint8, device=dev)
out = [torch.empty_like(rec) for _ in range(world)]
torch.cuda.synchronize(); t0 = time.perf_counter()
dist.all_gather(out, rec)
torch.cuda.synchronize()
print(f"rank {rank}/{world}: max = {t.item()}, all_gather of 52 MB per rank: {1e3 * (time.perf_counter() - t0):.2f} ms, "
      f"ok = {all(int(o[0]) == i for i, o in enumerate(out))}")
dist.barrier()
dist.destroy_process_group()
