"""One-rank RCCL self-check of the collectives the sampler uses (init, barrier, all_reduce MAX, all_gather)."""
import os, torch, torch.distributed as dist
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
lr = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(lr)
dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
x = torch.full((4, 1, 64000), float(dist.get_rank() + 1), device="cuda")
bufs = [torch.empty_like(x) for _ in range(dist.get_world_size())]
dist.all_gather(bufs, x)
t = torch.tensor([1.5], device="cuda", dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(bufs[dist.get_rank()], x) and t.item() == 1.5
print("rccl ok, world", dist.get_world_size())
dist.destroy_process_group()
