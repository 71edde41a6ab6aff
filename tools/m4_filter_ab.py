import os, sys, torch
sys.path.insert(0, os.getcwd())
import rayuela_jl_amd as rq
from rayuela_jl_amd import device as rqd, _lib
dev = "cuda"
def bench(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
g = torch.Generator(device=dev).manual_seed(1)
for (n, m, sub) in ((1_000_000, 4, 8), (1_000_000, 8, 16)):
    codes = rqd.synth_codes(n, m, seed=1234)
    centers = torch.randn((m, 256, sub), generator=g, device=dev) * 10
    for nq in (64, 1000, 4096, 10000):
        queries = torch.randn((nq, m * sub), generator=g, device=dev) * 10
        for K in (10, 100, 1000):
            out = (torch.empty((nq, K), dtype=torch.float32, device=dev), torch.empty((nq, K), dtype=torch.int32, device=dev))
            res = []
            for f in (1, 0):
                rq.set_tuning("SCAN_FILTER", f)
                res.append(bench(lambda: rqd.linscan(codes, centers, queries, K, out=out)))
            rq.set_tuning("SCAN_FILTER", 1)
            print("n=%d m=%d nq=%d K=%d  filter on %.4f ms  off %.4f ms  (%s)" % (n, m, nq, K, res[0], res[1], (_lib.lib().rq_last_scan_kernel() or b"").decode()), flush=True)
