import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import rayuela_jl_amd as rq, rayuela_jl_amd.synth as synth
from rayuela_jl_amd import device as rqd, _lib
n, d, m, h, K = 1_000_000, 128, 8, 256, 1000
X = synth.sift_like(n, d, seed=synth.SEED_BASE, ncentres=int(os.environ.get("NCENTRES", "1024")))

C = synth.codebooks(synth.sift_like(20_000, d, seed=synth.SEED_BASE, row0=3_100_000_000), m, h, seed=synth.SEED_CODEBOOK, iters=5, sample=20000)
B = torch.from_numpy(rq.quantize_pq_u8(X, C)).cuda(); cen = torch.from_numpy(np.stack(C)).cuda()
ob = rqd.order_rows(B)
for nq in [int(x) for x in os.environ.get("NQS", "512,2048,10000").split(",")]:
    Q = torch.from_numpy(synth.sift_like(nq, d, seed=synth.SEED_QUERY)).cuda()
    for name, base in (("arrival", B), ("ordered", ob)):
        rq.set_tuning("SCAN_ORDER", 0)
        rqd.linscan(base, cen, Q, K); rq.set_tuning("SCAN_STATS", 1); _lib.scan_stats(); rqd.linscan(base, cen, Q, K); torch.cuda.synchronize(); st = _lib.scan_stats(); rq.set_tuning("SCAN_STATS", 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): rqd.linscan(base, cen, Q, K)
        e1.record(); torch.cuda.synchronize()
        print(nq, name, "%.3f ms" % (e0.elapsed_time(e1) / 5), {k: st[k] for k in ("n_items", "n_fallbacks", "n_cuts")}, flush=True)
rq.set_tuning("SCAN_ORDER", 1)
