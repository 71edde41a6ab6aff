#!/usr/bin/env python3
"""Stress of the row order (sort + greedy balance): many base sizes around the balance's limits, both row widths, uniform / clumped /
duplicated / skewed / low-entropy codes; every result must be a permutation whose rows follow perm.  usage: python tools/order_stress.py [seconds]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
import rayuela_jl_amd as rq                      # noqa: E402
from rayuela_jl_amd import device as rqd, _lib   # noqa: E402

dev = torch.device("cuda", 0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
g = torch.Generator(device=dev).manual_seed(7)
t0 = time.time()
done = 0
sizes = [700_000, 790_000, 820_000, 1_000_000, 1_048_576, 1_500_001, 2_000_000, 2_999_999, 4_000_000, 5_300_000, 5_600_000, 8_000_000]
while time.time() - t0 < budget:
    for m in (8, 16):
        for n in sizes:
            if m == 16 and n > 5_000_000:
                continue
            for kind in ("uniform", "clumps", "dups", "skew", "lowbits", "onebyte"):
                if kind == "uniform":
                    c = torch.randint(0, 256, (n, m), generator=g, device=dev, dtype=torch.uint8)
                elif kind == "clumps":
                    pool = torch.randint(0, 256, (2048, m), generator=g, device=dev, dtype=torch.uint8)
                    c = pool[torch.randint(0, 2048, (n,), generator=g, device=dev)]
                    c[:, m - 1] = torch.randint(0, 256, (n,), generator=g, device=dev, dtype=torch.uint8)
                elif kind == "dups":
                    pool = torch.randint(0, 256, (37, m), generator=g, device=dev, dtype=torch.uint8)
                    c = pool[torch.randint(0, 37, (n,), generator=g, device=dev)]
                elif kind == "skew":
                    c = torch.randint(0, 256, (n, m), generator=g, device=dev, dtype=torch.uint8)
                    c[torch.randperm(n, generator=g, device=dev)[: n // 2]] = c[0].clone()
                elif kind == "lowbits":
                    c = torch.randint(0, 8, (n, m), generator=g, device=dev, dtype=torch.uint8)
                else:      # the free tables all equal: every column of a group takes the same byte
                    c = torch.randint(0, 256, (n, m), generator=g, device=dev, dtype=torch.uint8)
                    c[:, 4:] = 77
                c = c.contiguous()
                ob = rqd.order_rows(c)
                perm = ob.perm.long()
                ok = bool(torch.equal(torch.sort(perm).values, torch.arange(n, device=dev))) and bool(torch.equal(ob.codes[:, :m], c[perm]))
                done += 1
                if not ok:
                    print("FAILED", m, n, kind, flush=True)
                    sys.exit(1)
                del ob, c, perm
            if time.time() - t0 > budget:
                break
print("order_stress: %d orderings, all permutations (%.0f s)" % (done, time.time() - t0))
