"""World-size-1 RCCL smoke of the sharded path (single GPU box): ShardedIndex over NCCL + all_to_all."""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from rayuela_jl_amd.sharded import ShardedIndex
from rayuela_jl_amd import device as rqd
codes = rqd.synth_codes(200000, 8, 5)
cen = torch.randn(8, 256, 16, device="cuda"); q = torch.randn(64, 128, device="cuda")
r = ShardedIndex(codes, cen, 0).search(q, 100)
d0, i0 = rqd.linscan(codes, cen, q, 100)
print("RESULT nccl world=1 path ok:", torch.equal(r[0], d0), torch.equal(r[1], i0))
x = torch.arange(8, dtype=torch.int64, device="cuda"); y = torch.empty_like(x); dist.all_to_all_single(y, x)
g = [torch.empty_like(x)]; dist.all_gather(g, x)
gg = [torch.empty_like(x)]; dist.gather(x, gg, dst=0)
print("RESULT a2a", y.tolist(), "allgather", g[0].tolist(), "gather", gg[0].tolist())
dist.barrier(); dist.destroy_process_group()
