import os, sys, torch, torch.distributed as dist
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "prompt-cache_amd")]
from promptcache_amd import parallel
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29533"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
t = torch.ones(4, device="cuda"); dist.all_reduce(t); print("all_reduce", t.tolist(), flush=True)
sizes = [1000, 24, 4096 * 7 + 3]
mine, views = parallel.carve(sizes, torch.float16, "cuda")
mine.copy_(torch.randn(mine.numel(), device="cuda").half())
vb, _ = parallel.exchange_slabs(mine, [sizes, sizes], 0, 2, "cuda", rank_map=[0, 0])
torch.cuda.synchronize()
print("self loop equal:", all(bool((a == b).all()) for a, b in zip(vb[0], vb[1])), [v.numel() for v in vb[1]], flush=True)
dist.barrier(); dist.destroy_process_group(); print("done")
