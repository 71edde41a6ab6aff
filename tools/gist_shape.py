"""GIST1M-shape sanity and timing (d = 960, m = 8: sub = 120): encode, scan, and the training loops' phase clock.  GPU box."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayuela_jl_amd as rq
import rayuela_jl_amd.synth as synth
from rayuela_jl_amd import device as rqd, _lib
n, d, m, h = 200_000, 960, 8, 256
rng = np.random.default_rng(0)
X = (rng.standard_normal((n, d)) * (1.0 / (1 + np.arange(d) / 50.0))).astype(np.float32)
t0 = time.time(); C, B, err = rq.train_pq(X, m, h, 5, seed=1); t1 = time.time()
p = _lib.train_profile()
print("train_pq  niter 5: wall %.1f ms, loop %.2f ms/iter, error %.4f" % ((t1 - t0) * 1e3, p["loop_ms"] / max(1, p["iterations"]), err))
for fn, name in ((lambda: rq.train_opq(X, m, h, 5, "natural", seed=1), "train_opq"),):
    fn()
    t0 = time.time(); out = fn(); t1 = time.time()
    p = _lib.train_profile()
    rq.set_tuning("TRAIN_PROFILE", 1); fn(); f = _lib.train_profile(); rq.set_tuning("TRAIN_PROFILE", 0)
    it = max(1.0, f["iterations"])
    print(name, "niter 5: wall %.1f ms, loop %.2f ms/iter" % ((t1 - t0) * 1e3, p["loop_ms"] / max(1, p["iterations"])),
          {k: round(f[k] / it, 3) for k in ("qerror_ms", "gram_ms", "svd_ms", "rotate_ms", "update_centers_ms", "encode_ms", "reconstruct_ms") if f[k] > 0},
          "ns_steps %.1f host_polar %d" % (p["ns_steps"] / max(1, p["iterations"]), p["host_polar"]))
    Cq, Bq, R, obj = out
    print("   obj", obj, "R orth err %.2e" % np.abs(R @ R.T - np.eye(d)).max())
Xd = torch.from_numpy(X).cuda()
Ccat = torch.from_numpy(synth.cat_codebooks(C)).cuda()
for _ in range(3): codes = rqd.encode_pq(Xd, Ccat, m, h)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(5): codes = rqd.encode_pq(Xd, Ccat, m, h)
torch.cuda.synchronize(); print("encode_pq %d x %d: %.2f ms (%s)" % (n, d, (time.time() - t0) / 5 * 1e3, (_lib.lib().rq_last_encode_kernel() or b"").decode()))
