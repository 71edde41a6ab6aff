import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayuela_jl_amd as rq
from rayuela_jl_amd import device as rqd
for (n, d, m) in ((1_000_000, 128, 8), (1_000_000, 96, 16)):
    codes = torch.randint(0, 256, (n, m), dtype=torch.uint8, device="cuda")
    C = torch.randn(256 * d, device="cuda")
    out = torch.empty((n, d), device="cuda")
    for _ in range(3): rqd.reconstruct(codes, C, d, 256, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): rqd.reconstruct(codes, C, d, 256, out=out)
    e1.record(); torch.cuda.synchronize()
    print(n, d, m, "reconstruct %.3f ms" % (e0.elapsed_time(e1) / 10))
