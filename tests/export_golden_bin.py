#!/usr/bin/env python3
"""Raw little-endian mirrors of the encode / RVQ golden INPUTS (tests/golden/encode_*.npz, rvq_*.npz) under
tests/golden/bin/, so that a box with nothing but Julia can run julia/gen_encode_golden.jl on them.

    python tests/export_golden_bin.py          # rewrites tests/golden/bin/*.f32 and cases.txt

Layouts are the memory images Julia expects (column-major):  <case>.X.f32 = d x n matrix, <case>.C.f32 = the
m codebooks back to back (each sub_i x h, or d x h for RVQ), <case>.R.f32 = d x d.  cases.txt has one line per
case:  name kind n d m h has_R   (kind: pq | rvq).  tests/test_julia_golden.py checks the mirrors against the
.npz files and consumes the *_julia.i16 outputs when they are present."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
OUT = os.path.join(GOLD, "bin")
CASES = [("encode_sift_mini", "pq"), ("encode_deep_mini", "pq"), ("encode_uneven", "pq"), ("encode_h100", "pq"),
         ("rvq_sift_mini", "rvq"), ("rvq_deep_mini", "rvq")]


def main():
    os.makedirs(OUT, exist_ok=True)
    lines = []
    for name, kind in CASES:
        g = np.load(os.path.join(GOLD, name + ".npz"))
        X = np.ascontiguousarray(g["X"], dtype="<f4")
        n, d = X.shape
        if kind == "pq":
            m, h = int(g["m"]), int(g["h"])
            C = np.ascontiguousarray(g["C"], dtype="<f4")
        else:
            m, h = int(g["C"].shape[0]), int(g["C"].shape[1])
            C = np.ascontiguousarray(g["C"], dtype="<f4").reshape(-1)
        X.tofile(os.path.join(OUT, name + ".X.f32"))
        C.tofile(os.path.join(OUT, name + ".C.f32"))
        has_R = int("R" in g.files)
        if has_R:
            np.ascontiguousarray(g["R"], dtype="<f4").tofile(os.path.join(OUT, name + ".R.f32"))
        lines.append("%s %s %d %d %d %d %d" % (name, kind, n, d, m, h, has_R))
    with open(os.path.join(OUT, "cases.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("wrote %d cases to %s" % (len(lines), OUT))


if __name__ == "__main__":
    main()
