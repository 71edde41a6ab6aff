import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import rayuela_jl_amd as rq
import rayuela_jl_amd.synth as synth
import rayuela_jl_amd.synth_torch as st
from rayuela_jl_amd import device as rqd
dev = torch.device("cuda", 0)
n, h, d, m = 1_000_000, 256, 128, 8
X = torch.cat([st.sift_like(250_000, d, seed=synth.SEED_BASE, ncentres=65536, row0=o, device=dev) for o in range(0, n, 250_000)], 0)
C = synth.codebooks(X[:20000].cpu().numpy(), m, h, seed=synth.SEED_CODEBOOK, iters=3, sample=20000)
Ccat = torch.from_numpy(synth.cat_codebooks(C)).to(dev)
rq.set_tuning("ENC_SPLIT", 0)
ref = rqd.encode_pq(X, Ccat, m, h).clone()
rq.set_tuning("ENC_SPLIT", 1)
for rep in range(3):
    got = rqd.encode_pq(X, Ccat, m, h)
    torch.cuda.synchronize()
    idx = (got != ref).nonzero().cpu().numpy()
    print("rep", rep, "diffs", len(idx))
    Xh = X.cpu().numpy()
    for (row, i) in idx[:12]:
        sub = d // m
        x = Xh[row, i * sub:(i + 1) * sub].astype(np.float64)
        c = C[i].astype(np.float64)
        dist = ((c - x) ** 2).sum(1)
        a, b = int(ref[row, i]), int(got[row, i])
        srt = np.argsort(dist)[:3]
        print("  row %d (row%%32=%d) subq %d: ref %d (d=%.3f) split %d (d=%.3f); best3 %s %s" % (row, row % 32, i, a, dist[a], b, dist[b], srt, dist[srt]))
