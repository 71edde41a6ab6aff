#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: ADC queries/sec + encode vectors/sec at SIFT1M shape.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload pq|opq|deep] [--k 1000] [--nq 10000]

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in
HBM: (a) ADC scan + exact top-k of nq queries over the rank's 1e6-row shard (linscan_pq, the headline
`value`), and, timed in its own bracket, (b) quantize_pq of the rank's 1e6 x d base (reported under
"encode").  N > 1 ranks (one per GPU, RCCL): weak scaling -- every rank keeps a 1e6-row shard, so the
base grows to N x 1e6 rows; the step then includes the all_to_all exchange of per-shard top-k keys, the
merge and the gather to rank 0.  `value` counts one unit per (query x 1e6-row shard) processed, which at
N = 1 is exactly queries/s against the SIFT1M-shape base.

Rank 0 prints ONE JSON line (contract in the task description) with two extra objects:
  roofline      dominant kernel (ADC scan): algorithmic bytes nq*n*m per launch / measured launch time
                vs the 8 TB/s HBM3E spec; LDS-gather bound quoted beside it
  cpu_baseline  the reference's own deps/src/linscan_aqd.cpp (oracle/_ref, built by oracle/Makefile)
                timed on this box's host cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_PEAK_TFLOPS = 157.3     # f32 MFMA == f32 vector peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="pq", choices=["pq", "opq", "deep"])
    ap.add_argument("--n", type=int, default=1_000_000, help="base rows per GPU")
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def make_data(n, nq, d, kind, rank, device):
    """SIFT-like / Deep-like synthetic vectors generated on the device (seeded torch generator)."""
    g = torch.Generator(device=device).manual_seed(1234 + 7919 * rank)
    gq = torch.Generator(device=device).manual_seed(4321)
    gc = torch.Generator(device=device).manual_seed(99)
    if kind == "sift":
        ncent = 65536   # ~15 base vectors per centre at 1e6 rows: near-duplicates like SIFT, but resolvable
        cent = torch.randint(0, 128, (ncent, d), generator=gc, device=device).float()

        def gen(rows, gen_):
            cid = torch.randint(0, ncent, (rows,), generator=gen_, device=device)
            noise = torch.randint(-16, 17, (rows, d, 4), generator=gen_, device=device).sum(-1).float()
            return (cent[cid] + noise).clamp_(0, 255).contiguous()
    else:
        def gen(rows, gen_):
            v = torch.rand((rows, d, 4), generator=gen_, device=device).sum(-1) - 2.0
            return (v / v.norm(dim=1, keepdim=True)).float().contiguous()
    X = torch.cat([gen(min(250_000, n - a), g) for a in range(0, n, 250_000)], 0)
    Q = gen(nq, gq)
    S = gen(20_000, gc)   # codebook training sample: identical on every rank
    return X, Q, S


def timed(fn, steps, warmup, barrier):
    for _ in range(warmup):
        fn()
    barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    barrier()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3
    return e0.elapsed_time(e1), wall_ms


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    def barrier():
        if world > 1:
            dist.barrier()

    import rayuela_jl_amd as rq
    import rayuela_jl_amd.synth as synth
    from rayuela_jl_amd import device as rqd
    from rayuela_jl_amd.sharded import ShardedIndex

    if a.workload == "deep":
        d, m, kind, name = 96, 16, "deep", "Deep1M-shape OPQ d=96 m=16 h=256"
    else:
        d, m, kind = 128, 8, "sift"
        name = "SIFT1M-shape %s m=8 h=256" % ("OPQ" if a.workload == "opq" else "PQ")
    use_R = a.workload in ("opq", "deep")
    h, n, nq, K = 256, a.n, a.nq, a.k

    X, Q, S = make_data(n, nq, d, kind, rank, device)
    R = torch.from_numpy(synth.rotation(d)).to(device) if use_R else None
    # codebooks: k-means on the (rotated, for OPQ) training sample -- harness side, untimed
    S_train = rqd.rotate_T(R, S) if use_R else S
    C = synth.codebooks(S_train.cpu().numpy(), m, h, seed=synth.SEED_CODEBOOK, iters=5, sample=20000)
    Ccat = torch.from_numpy(synth.cat_codebooks(C)).to(device)
    centers = torch.from_numpy(np.stack(C)).to(device)

    # ---- (b) encode: quantize_pq / quantize_opq of the resident base -----------------------------------
    codes = torch.empty((n, m), dtype=torch.uint8, device=device)
    if use_R:
        enc = lambda: rqd.encode_opq(X, R, Ccat, m, h, out=codes)   # noqa: E731
    else:
        enc = lambda: rqd.encode_pq(X, Ccat, m, h, out=codes)       # noqa: E731
    enc_ms, _ = timed(enc, a.steps, a.warmup, barrier)

    # ---- (a) ADC scan + top-k -------------------------------------------------------------------------
    Qs = rqd.rotate_T(R, Q) if use_R else Q     # linscan_opq rotates the queries first (src/Linscan.jl:102)
    if world == 1:
        out = (torch.empty((nq, K), dtype=torch.float32, device=device),
               torch.empty((nq, K), dtype=torch.int32, device=device))
        scan = lambda: rqd.linscan(codes, centers, Qs, K, out=out)   # noqa: E731
    else:
        ix = ShardedIndex(codes, centers, id_offset=rank * n)
        res = {}

        def scan():
            res["r"] = ix.search(Qs, K)
    scan_ms, scan_wall = timed(scan, a.steps, a.warmup, barrier)

    t = torch.tensor([scan_ms, enc_ms, scan_wall], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    scan_ms, enc_ms, scan_wall = [float(x) for x in t.tolist()]
    ms_step = scan_ms / a.steps
    enc_ms_step = enc_ms / a.steps
    qps = world * nq / (ms_step * 1e-3)
    vps = world * n / (enc_ms_step * 1e-3)

    if rank != 0:
        barrier()
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- roofline: algorithmic bytes per launch / measured launch time ---------------------
    # HBM/fabric bytes per launch from the PMC passes of the same shape (rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE, separate runs; FETCH doubled per MI355X_MICROARCH.md section HBM), committed under profiles/
    traffic_bytes, traffic_src = None, None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r1_traffic.json")))
        key = "adc_scan_kernel<%d> n=%d nq=%d k=%d" % (m, n, nq, K)
        if key in tj:
            traffic_bytes = 2.0 * tj[key]["FETCH_SIZE_KiB"] * 1024 + tj[key]["WRITE_SIZE_KiB"] * 1024
            traffic_src = "profiles/r1_traffic.json (%s)" % tj[key]["source"]
    except Exception:
        pass
    scan_bytes = float(nq) * n * m                       # n*m code bytes per query
    achieved = scan_bytes / (ms_step * 1e-3) / 1e9      # GB/s, per GPU
    roof = {"bound": "hbm", "kernel": "adc_scan_kernel<%d>" % m, "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic_bytes, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": scan_bytes,
            "note": "codes are read once per 8-query group, so algorithmic GB/s may exceed HBM; the binding "
                    "resource is the LDS gather (ds_read_b128, 4 queries per gather)",
            "lds_gathers_per_s": round(float(nq) * n * m / 4.0 / (ms_step * 1e-3), 1)}
    enc_flops = 2.0 * d * h * n
    enc_roof = {"bound": "mfma", "kernel": "encode_pq_kernel", "achieved": round(enc_flops / (enc_ms_step * 1e-3) / 1e12, 2),
                "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(enc_flops / (enc_ms_step * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4), "traffic": None,
                "hbm_GBps": round((4.0 * d + m) * n / (enc_ms_step * 1e-3) / 1e9, 1)}
    if use_R:
        enc_roof["note"] = "includes the R'X rotation kernel (2*d*d flop/vector more, not counted in achieved)"

    # ---- recall (sanity + parity: ids are bit-exact, so recall is identical by construction) ---------------
    recall = None
    if world == 1:
        nrec = min(nq, 1000)
        best = torch.full((nrec,), float("inf"), device=device)
        arg = torch.zeros((nrec,), dtype=torch.long, device=device)
        Qf = Q[:nrec].double()
        for a0 in range(0, n, 100_000):
            Xc = X[a0:a0 + 100_000].double()
            dd = (Qf * Qf).sum(1)[:, None] - 2.0 * Qf @ Xc.T + (Xc * Xc).sum(1)[None, :]
            v, i = dd.min(1)
            upd = v < best
            best = torch.where(upd, v, best)
            arg = torch.where(upd, i + a0, arg)
        ids = out[1][:nrec].long() & 0xFFFFFFFF
        rec = rq.eval_recall(arg.cpu().numpy(), ids.cpu().numpy(), K, verbose=False)
        recall = {"r@1": float(rec[0]), "r@10": float(rec[min(9, K - 1)]), "r@100": float(rec[min(99, K - 1)]),
                  "r@%d" % K: float(rec[K - 1]), "queries": nrec}

    # ---- CPU baseline on this box's host cores (rank 0, N = 1 only) -----------------------------------------
    cpu = None
    if world == 1 and not a.no_cpu:
        from oracle import oracle
        codes_h = codes.cpu().numpy()
        cen_h = centers.cpu().numpy()
        Q_h = Qs.cpu().numpy()
        cores = os.cpu_count() or 1
        use_ref = oracle.ref_available()
        fn = oracle.ref_linscan_aqd_query if use_ref else oracle.linscan_aqd_query
        s0 = max(1, min(nq, cores))
        t0 = time.perf_counter()
        fn(codes_h, cen_h, Q_h[:s0], K)
        dt0 = time.perf_counter() - t0
        s1 = int(min(nq, max(s0, a.cpu_seconds / max(dt0 / s0, 1e-6))))
        s1 = max(s0, (s1 // s0) * s0)
        t0 = time.perf_counter()
        d_cpu, i_cpu = fn(codes_h, cen_h, Q_h[:s1], K)
        dt1 = time.perf_counter() - t0
        same = bool(np.array_equal(i_cpu, out[1][:s1].cpu().numpy().view(np.uint32)) and
                    np.array_equal(d_cpu.view(np.uint32), out[0][:s1].cpu().numpy().view(np.uint32)))
        ne = min(n, 1_000_000)
        Xh = (rqd.rotate_T(R, X[:ne]) if use_R else X[:ne]).cpu().numpy()
        t0 = time.perf_counter()
        c_cpu = oracle.encode_pq(Xh, synth.cat_codebooks(C), m, h)
        dte = time.perf_counter() - t0
        cpu = {"value": round(s1 / dt1, 2), "unit": "queries/s", "cores": cores,
               "kind": "reference" if use_ref else "port",
               "sample": "%d of the %d queries, full 1e6-row base, k=%d (%.1f s); deps/src/linscan_aqd.cpp built "
                         "g++ -O3 -fopenmp as in deps/build.jl:23" % (s1, nq, K, dt1),
               "gpu_matches_cpu_bit_exact": same,
               "encode": {"value": round(ne / dte, 1), "unit": "vectors/s", "kind": "port", "cores": oracle.num_threads(),
                          "sample": "%d vectors (%.2f s), oracle/rq_oracle.c" % (ne, dte),
                          "codes_match": bool(np.array_equal(c_cpu, codes[:ne].cpu().numpy()))}}

    line = {
        "metric": "ADC queries/sec (linscan_pq, exact top-%d)" % K if not use_R else "ADC queries/sec (linscan_opq, exact top-%d)" % K,
        "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": name + " encode + ADC linscan", "n_base_per_gpu": n, "n_base_total": n * world,
                   "nq": nq, "k": K, "d": d, "m": m, "h": h,
                   "unit_of_value": "one query scanned against one 1e6-row shard; = queries/s at 1 GPU",
                   "parallelism": "row-sharded base x%d, all_to_all top-k exchange + merge" % world if world > 1 else "single GPU"},
        "encode": {"metric": "encode vectors/sec (%s)" % ("quantize_opq" if use_R else "quantize_pq"),
                   "value": round(vps, 1), "unit": "vectors/s", "ms_per_step": round(enc_ms_step, 4), "roofline": enc_roof},
        "roofline": roof,
        "cpu_baseline": cpu,
        "recall": recall,
        "wall_ms_per_step": round(scan_wall / a.steps, 4),
    }
    print(json.dumps(line))
    sys.stdout.flush()
    barrier()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
