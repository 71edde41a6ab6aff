#!/usr/bin/env python3
"""rq_index_search with 1 / 2 / 4 / 8 LOGICAL shards on device 0 over the same rows (Next #1d): what the ONE
host thread that issues every shard's scan, the exchange and the merge (csrc/rq_index.hip) costs per extra shard.
On one GPU the shards' kernels share the device, so the kernel time is (roughly) constant and the difference between the
rows of the table is launch + exchange + merge overhead.  Prints a markdown table."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import rayuela_jl_amd as rq                      # noqa: E402
import rayuela_jl_amd.synth as synth             # noqa: E402


def main():
    d, m, nq, k = 128, 8, 1024, 100
    # bench.py's sift1b inputs: codebooks and queries from sift-like vectors, base = hash-generated code bytes
    S = synth.sift_like(20_000, d, seed=synth.SEED_BASE, ncentres=65536, row0=3_100_000_000)
    C = synth.codebooks(S, m, 256, seed=synth.SEED_CODEBOOK, iters=5, sample=20000)
    Q = synth.sift_like(nq, d, seed=synth.SEED_BASE, ncentres=65536, row0=3_000_000_000)
    print("| rows | shards | exchange | ms per search (best of 7) | per extra shard (us) |")
    print("|---|---|---|---|---|")
    for n in (100_000, 16_000_000, 125_000_000):
        base = None
        ref = None
        for P in (1, 2, 4, 8):
            with rq.Index(C, d, devices=[0] * P) as ix:
                ix.set_codes_synth(n, synth.SEED_BASE)
                best = 1e30
                for it in range(8):
                    t0 = time.perf_counter()
                    dists, ids = ix.search(Q, k, id_base=0)
                    dt = time.perf_counter() - t0
                    if it:
                        best = min(best, dt)
                ex = ix.info()["exchange"]
            if ref is None:
                ref = (dists.copy(), ids.copy())
            else:
                assert np.array_equal(ids, ref[1]) and np.array_equal(dists.view(np.uint32), ref[0].view(np.uint32))
            base = best if base is None else base
            print("| %d | %d | %s | %.3f | %s |" % (n, P, ex, best * 1e3, "-" if P == 1 else "%.0f" % ((best - base) * 1e6 / (P - 1))))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
