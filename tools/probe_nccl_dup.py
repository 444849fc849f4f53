"""Measurement aid: can several ranks share cuda:0 under RCCL on this box?"""
import os, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
r, w = dist.get_rank(), dist.get_world_size()
t = torch.full((4,), float(r + 1), dtype=torch.float64, device="cuda")
dist.all_reduce(t)
send = torch.arange(6, dtype=torch.float64, device="cuda").reshape(3, 2) + 10 * r
splits = [1, 2] if r == 0 else [2, 1]
recv = torch.empty((3, 2), dtype=torch.float64, device="cuda")
dist.all_to_all_single(recv, send, splits, splits)
torch.cuda.synchronize()
print(r, "ok", t.tolist(), recv.tolist(), flush=True)
dist.destroy_process_group()
