"""Rotation kernel timing (R'X at 1e6 x d).  usage: python tools/rotate_perf.py  (GPU box; RAYUELA_HIP_LIB picks the build)"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayuela_jl_amd as rq
import rayuela_jl_amd.synth as synth
from rayuela_jl_amd import device as rqd
for d in (128, 96, 960, 320):
    n = 1_000_000 if d <= 128 else 200_000
    X = torch.randn((n, d), device="cuda")
    R = torch.from_numpy(synth.rotation(d)).cuda()
    out = torch.empty_like(X)
    for _ in range(5): rqd.rotate_T(R, X, out=out)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): rqd.rotate_T(R, X, out=out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print("rotate n=%d d=%d  %.4f ms  %.1f TF  %.2f TB/s" % (n, d, best, 2.0 * d * d * n / best / 1e9, 8.0 * d * n / best / 1e9))
