#!/usr/bin/env python3
"""Encode timing on the bench's data (development aid): ms per 1e6 vectors for the library RAYUELA_HIP_LIB points at, and
how many codes differ from a reference library's (RQ_REF_LIB, default: the shipped one) -- run once per library."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayuela_jl_amd as rq
import rayuela_jl_amd.synth as synth
import rayuela_jl_amd.synth_torch as st
from rayuela_jl_amd import device as rqd

dev = torch.device("cuda", 0)
save = os.environ.get("ENC_SAVE")        # directory: codes of this library are saved / compared with what is there


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for kind in sys.argv[1:] or ["sift", "deep"]:
    n, h = 1_000_000, 256
    if kind == "sift":
        d, m = 128, 8
        X = torch.cat([st.sift_like(250_000, d, seed=synth.SEED_BASE, ncentres=65536, row0=o, device=dev) for o in range(0, n, 250_000)], 0)
    else:
        d, m = 96, 16
        X = torch.cat([st.deep_like(250_000, d, seed=synth.SEED_BASE, row0=o, device=dev) for o in range(0, n, 250_000)], 0)
    C = synth.codebooks(X[:20000].cpu().numpy(), m, h, seed=synth.SEED_CODEBOOK, iters=3, sample=20000)
    Ccat = torch.from_numpy(synth.cat_codebooks(C)).to(dev)
    out = torch.empty((n, m), dtype=torch.uint8, device=dev)
    rq.set_tuning("ENC_STATS", 1)        # share of (vector, sub-quantizer) pairs the filter leaves to the exact pass
    got = rqd.encode_pq(X, Ccat, m, h).cpu().numpy()
    import ctypes as C
    from rayuela_jl_amd import _lib
    stt = (C.c_uint64 * 2)()
    _lib.lib().rq_last_encode_stats(C.cast(stt, C.c_void_p))
    rq.set_tuning("ENC_STATS", 0)
    print("%s: %d of %d pairs to the exact pass (%.3f %%)" % (kind, stt[1], stt[0], 100.0 * stt[1] / max(1, stt[0])), flush=True)
    t1 = bench(lambda: rqd.encode_pq(X, Ccat, m, h, out=out))
    msg = ""
    if save:
        f = os.path.join(save, "codes_%s.npy" % kind)
        if os.path.exists(f):
            msg = "  differing from %s: %d of %d" % (f, int((np.load(f) != got).sum()), n * m)
        else:
            os.makedirs(save, exist_ok=True)
            np.save(f, got)
            msg = "  (saved %s)" % f
    print("%s %s %.4f ms%s" % (os.path.basename(os.environ.get("RAYUELA_HIP_LIB", "shipped")), kind, t1, msg), flush=True)
