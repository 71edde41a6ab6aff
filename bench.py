#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: ADC queries/sec + encode vectors/sec.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload auto|pq|opq|deep|sift1b] [--k K] [--nq NQ]

Workloads (BASELINE.json `configs`):
  pq      SIFT1M-shape PQ  m=8 h=256: quantize_pq + linscan_pq            (config 2; default at N = 1)
  opq     SIFT1M-shape OPQ m=8 h=256: rotation + encode + ADC               (config 3)
  deep    Deep1M-shape OPQ d=96 m=16 h=256                                   (config 4)
  sift1b  SIFT1B-shape base, 1e9 x 8 uint8 codes generated on the devices, nq=1024, k=100, rows sharded over
          the N GPUs, per-shard top-k gathered over xGMI (RCCL) and merged    (config 5; default at N > 1)
  train_opq / train_pq   SURVEY 8f rank 1 (src/OPQ.jl:49-139, src/PQ.jl:68-99) on the SIFT1M-shape base through the C ABI
          (rq_train_opq / rq_train_pq): a step = ONE training iteration; --steps = niter (the demos use 25); N = 1 only

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM: the ADC scan +
exact top-k of nq queries over the WHOLE base (`value` = true queries/s against the whole base), and -- timed in
its own bracket, reported under "encode" -- quantize_pq of the base.  N > 1 is STRONG scaling: the base is
fixed and its rows are split over the ranks; a step then includes the exchange of per-shard top-k keys, the
merge and the gather to rank 0.

Launch: `python bench.py --gpus N` starts N ranks itself (python -m torch.distributed.run, one process per GPU,
backend nccl == RCCL); started under torch.distributed.run it uses the ranks it is given.  `--inproc` runs the
N-GPU search inside ONE process through the library's multi-device index (rq_index_create_sharded: what a Julia
session gets) instead.

Rank 0 prints ONE JSON line (contract in the task description) with two extra objects:
  roofline      the dominant kernel (adc_scan_kernel) against the resource that binds it -- the LDS gather pipe:
                algorithmic bytes nq*n*m per launch / kernel time vs 256 CU x 256 B/clk x 2.4 GHz; the HBM side
                (the same bytes vs 8 TB/s, and the PMC-measured traffic) is reported next to it
  cpu_baseline  the reference's own deps/src/linscan_aqd.cpp (oracle/_ref, built by oracle/Makefile) timed on
                this box's host cores on a bounded sample of the same workload
and, because N = 1 and N > 1 default to DIFFERENT BASELINE configs (2 and 5), the anchors that make a series of
`--gpus 1, 2, 4, 8` runs comparable: the default N = 1 line carries `scale_anchor_1gpu` (config 5's workload on this one
GPU: ~0.6 s extra), every N > 1 line carries `same_workload_1gpu` (rank 0 alone on the whole base).  Efficiency over N
is value(N) / (N x that single-GPU value) -- not value(N) / (N x value(1)).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_PEAK_TFLOPS = 157.3     # f32 MFMA == f32 vector peak
NUM_CU, LDS_B_PER_CLK, CLK_GHZ = 256, 256, 2.4     # MI355X_MICROARCH.md section LDS: 256 B/clk/CU for ds_read_b128
LDS_PEAK_GBS = NUM_CU * LDS_B_PER_CLK * CLK_GHZ    # 157 286 GB/s conflict-free


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="auto", choices=["auto", "pq", "opq", "deep", "sift1b", "train_opq", "train_pq"])
    ap.add_argument("--rows", dest="n", type=int, default=0, help="rows of the WHOLE base (default 1e6; sift1b: 1e9)")
    ap.add_argument("--nq", type=int, default=0, help="queries per step (default 10000; sift1b: 1024)")
    ap.add_argument("--k", type=int, default=0, help="neighbours (default 1000; sift1b: 100)")
    ap.add_argument("--inproc", action="store_true", help="N GPUs from one process via rq_index_create_sharded")
    ap.add_argument("--devices", default="", help="--inproc only: explicit device list, e.g. 0,0,0,0 = four LOGICAL shards on "
                    "device 0 (functional check of the sharded index on a one-GPU box; --gpus is set to its length)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-ref1", action="store_true", help="skip the single-GPU timing of the N > 1 workload (N > 1: same_workload_1gpu; default N = 1 run: scale_anchor_1gpu)")
    ap.add_argument("--no-host", action="store_true", help="skip the host-pointer (PCIe-inclusive) timing")
    ap.add_argument("--no-ab", action="store_true", help="skip the prepared-base / arrival-order comparison runs (profiling: every "
                    "scan launch of the run is then the headline's kernel on ordered rows)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_spawn(a):
    """Plain `python bench.py --gpus N`: become N ranks, one per GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def timed(fn, steps, warmup, barrier):
    import torch
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


def lib_build():
    """the 12-hex-digit id of the kernel sources the loaded library was built from (rq_version())"""
    from rayuela_jl_amd import _lib
    v = (_lib.lib().rq_version() or b"").decode()
    return v.split("build ")[-1] if "build " in v else "?"


def load_traffic(kernel_key):
    """HBM/fabric bytes per launch of `kernel_key` from THIS round's committed PMC passes (profiles/r6_traffic.json: rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE, separate runs; FETCH doubled per MI355X_MICROARCH.md section HBM).  The figure is a REPLAY
    of a profiled run, so it is bound to what it was measured on: the entry names the kernel instantiation (as rocprofv3 and
    rq_last_scan_kernel() spell it) and the library build id (rq_version()); if either differs from what just ran, traffic
    is null and the reason is reported instead."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r6_traffic.json")))
    except Exception:
        return None, "no profiles/r6_traffic.json"
    e = tj.get(kernel_key)
    if e is None:
        return None, "profiles/r6_traffic.json has no entry for this shape"
    from rayuela_jl_amd import _lib
    ran = (_lib.lib().rq_last_scan_kernel() or b"").decode()
    build = lib_build()
    if e.get("build") != build or e.get("kernel") != ran:
        return None, ("not replayed: profiles/r6_traffic.json was measured on %s of build %s, this run launched %s of build %s"
                      % (e.get("kernel"), e.get("build"), ran, build))
    return (2.0 * e["FETCH_SIZE_KiB"] * 1024 + e["WRITE_SIZE_KiB"] * 1024,
            "profiles/r6_traffic.json (%s; kernel %s, build %s: the ones this run used)" % (e.get("source", ""), ran, build))


def load_encode_counters(kernel, sub):
    """per-launch instruction counters of the encode kernels from this round's PMC passes (profiles/r6_encode_counters.json), bound
    to the library build like the traffic figures; None when they do not describe the library that just ran"""
    try:
        ej = json.load(open(os.path.join(ROOT, "profiles", "r6_encode_counters.json")))
    except Exception:
        return None
    e = ej.get("%s sub=%d" % (kernel, sub))
    return e if e is not None and e.get("build") == lib_build() else None


def scan_roofline(m, n_local, nq, K, kernel_ms):
    """The ADC scan kernel against the resource that binds it, the LDS gather pipe.

    Algorithmic bytes per launch = nq * n * m (one table entry looked up per code byte per query).
    Since round 2 the hot loop looks a ONE-byte lower bound up for (almost) every (query, row, sub-quantizer) --
    8 queries per ds_read_b64 gather -- so nq*n*m is also the number of table bytes the LDS has to deliver, and
    the roof is the conflict-free LDS rate, 256 CU x 256 B/clk x 2.4 GHz.  `frac` is what is left of it by bank conflicts
    (47 % of the LDS cycles on the ordered 1e6-row base; 63 % in arrival order), by VALU issue that does not overlap the
    LDS pipe (66 % / 63 % busy), by the exact re-evaluation of surviving rows and by the top-k finish (profiles/).
    `f32_table_roof` keeps round 1's yardstick (4-byte entries, 4 queries per ds_read_b128) for continuity.
    HBM: the same nq*n*m code bytes are shared by the 8 queries of a group and mostly served by L2/MALL, so
    the PMC traffic is what reaches the fabric."""
    t = kernel_ms * 1e-3
    code_bytes = float(nq) * n_local * m
    achieved = code_bytes / t / 1e9
    traffic, src = load_traffic("adc_scan_kernel<%d> n=%d nq=%d k=%d" % (m, n_local, nq, K))
    hbm = {"algorithmic_GBps": round(achieved, 1), "peak_GBps": HBM_PEAK_GBS,
           "effective_frac": round(achieved / HBM_PEAK_GBS, 4),
           "traffic_bytes_per_launch": traffic, "traffic_source": src,
           "traffic_GBps": None if traffic is None else round(traffic / t / 1e9, 1),
           "traffic_frac_of_peak": None if traffic is None else round(traffic / t / 1e9 / HBM_PEAK_GBS, 4)}
    return {"bound": "lds", "kernel": "adc_scan_kernel<%d>" % m, "achieved": round(achieved, 1), "peak": round(LDS_PEAK_GBS, 1),
            "unit": "GB/s", "frac": round(achieved / LDS_PEAK_GBS, 4), "traffic": traffic,
            "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": code_bytes,
            "definition": "nq*n*m one-byte table look-ups per launch / kernel time vs %d CU x %d B/clk x %.1f GHz "
                          "(conflict-free LDS gather rate)" % (NUM_CU, LDS_B_PER_CLK, CLK_GHZ),
            "f32_table_roof": {"achieved_GBps": round(4.0 * achieved, 1), "frac": round(4.0 * achieved / LDS_PEAK_GBS, 4),
                               "note": "round-1 formula: 4-byte entries (nq*n*m*4 B); 0.31 in round 1"},
            "hbm": hbm}


def train_bench(a):
    """`--workload train_opq|train_pq`: one step = one iteration of the training loop on the resident SIFT1M-shape base.
    ms_per_step = the library's own wall clock over the iteration loop / iterations (X upload, initialisation and the
    download of the results are outside it and reported next to it); the phase split comes from a second, profiled call
    (TRAIN_PROFILE = 1: every phase between device synchronisations)."""
    import numpy as np
    import torch
    import rayuela_jl_amd as rq
    import rayuela_jl_amd.synth as synth
    import rayuela_jl_amd.synth_torch as st
    from rayuela_jl_amd import _lib
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    opq = a.workload == "train_opq"
    d, m, h = 128, 8, 256
    n = a.n or 1_000_000
    niter = max(1, a.steps)
    X = torch.cat([st.sift_like(min(250_000, n - o), d, seed=synth.SEED_BASE, ncentres=65536, row0=o, device=device)
                   for o in range(0, n, 250_000)], 0).cpu().numpy()

    def run(it):
        if opq:
            return rq.train_opq(X, m, h, it, "natural", seed=7)
        return rq.train_pq(X, m, h, it, seed=7)

    if a.warmup > 0:
        run(niter)          # one whole call: allocations, kernel images and the clocks are warm for the timed one
    t0 = time.perf_counter()
    out = run(niter)
    wall = time.perf_counter() - t0
    prof = _lib.train_profile()
    iters = max(1.0, prof["iterations"])
    ms_step = prof["loop_ms"] / iters
    rq.set_tuning("TRAIN_PROFILE", 1)
    try:
        run(niter)
        fine = _lib.train_profile()
    finally:
        rq.set_tuning("TRAIN_PROFILE", 0)
    fi = max(1.0, fine["iterations"])
    per_iter = {k: round(fine[k] / fi, 4) for k in ("qerror_ms", "gram_ms", "svd_ms", "rotate_ms", "update_centers_ms", "encode_ms",
                                                    "reconstruct_ms", "converge_ms") if fine[k] > 0}
    # (train_pq: qerror / reconstruct run ONCE after the loop; their entries are that one call's time / iterations)
    # every device phase of the LOOP against its own roof (algorithmic bytes / flops per iteration, SURVEY 8d's style).  Phases
    # that run once after the loop (train_pq: the final qerror / reconstruct) are not loop phases and get no roof.
    GB = 1e9
    once = set() if opq else {"qerror_ms", "reconstruct_ms"}
    shapes = {"rotate_ms": ("mfma", 2.0 * d * d * n), "gram_ms": ("mfma", 2.0 * d * d * n), "encode_ms": ("mfma", 2.0 * d * h * n),
              "update_centers_ms": ("hbm", (4.0 * d + m) * n), "reconstruct_ms": ("hbm", (4.0 * d + m) * n),
              # qerror: X and the n x d reconstruction, or X and the codes when CB is gathered inside the kernel (no reconstruct phase)
              "qerror_ms": ("hbm", (8.0 * d if "reconstruct_ms" in per_iter else 4.0 * d + m) * n)}
    roofs = {}
    for k, (bound, work) in shapes.items():
        if k not in per_iter or k in once:
            continue
        t = per_iter[k] * 1e-3
        if bound == "mfma":
            ach = work / t / 1e12
            roofs[k[:-3]] = {"bound": "mfma", "achieved": round(ach, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / FP32_PEAK_TFLOPS, 4)}
        else:
            ach = work / t / GB
            roofs[k[:-3]] = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}
    if "update_centers" in roofs:
        # the segment sums run as one-hot products on the bf16 matrix cores (csrc/rq_train.hip: centers_mfma_kernel): 4 MFMAs
        # of 16 x 16 x 32 per (32 rows, 16-dimension block, 16-code tile); the HBM roof above is the algorithmic one
        units = sum((sz + 15) // 16 for sz in [d // m + (1 if i < d % m else 0) for i in range(m)])
        mf = 4 * 2.0 * 16 * 16 * 32 * ((h + 15) // 16) * units * (n / 32.0) / (per_iter["update_centers_ms"] * 1e-3) / 1e12
        roofs["update_centers"]["bf16_mfma"] = {"issued_TFLOPs": round(mf, 1), "peak": 2500.0, "frac": round(mf / 2500.0, 4)}
    if "encode" in roofs:
        # the assignment step is the split encode kernel: its products run as a FILTER on the bf16 matrix cores (3 K = 16 MFMAs per
        # 32 x 32 tile at sub = 16), so it is priced by the bf16 work it issues against the bf16 peak; the f32-equivalent figure
        # (2 d h flop per vector against the f32 matrix peak, SURVEY 8d's yardstick) rides along and may pass 1
        bf = 2.0 * 16 * 256 * (2 if d // m <= 8 else 3) * m * n / (per_iter["encode_ms"] * 1e-3) / 1e12
        f32eq = roofs["encode"]
        roofs["encode"] = {"bound": "mfma", "achieved": round(bf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(bf / 2500.0, 4),
                           "dtype": "bf16 (filter) + f32 VALU re-evaluation of the candidates; the VALU epilogue binds",
                           "f32_equivalent": {"achieved": f32eq["achieved"], "peak": f32eq["peak"], "frac": f32eq["frac"]}}
    dev_phases = {k: v for k, v in per_iter.items() if k not in ("svd_ms", "converge_ms") and k not in once}
    dom = max(dev_phases, key=dev_phases.get)[:-3] if dev_phases else None
    roof = dict(roofs.get(dom, {}), kernel=dom, traffic=None, per_phase=roofs,
                note="the dominant throughput phase of an iteration; svd_ms is the d x d polar factor on the device (scaled "
                     "Newton-Schulz, %.1f steps of two grid-synchronised d^3 products in double per iteration: latency-bound, "
                     "no roof applies)" % (prof["ns_steps"] / iters)) if dom else None
    # CPU baseline: the numpy/oracle restatement of the same loop (oracle/train_oracle.py) on a bounded sample
    cpu = None
    if not a.no_cpu:
        from oracle import oracle, train_oracle as to
        ns = min(n, 100_000)
        Xs = X[:ns]
        off = to.offsets(d, m)
        rngc = np.random.default_rng(0)
        C0 = [Xs[rngc.choice(ns, h, replace=False)][:, off[i]:off[i + 1]].copy() for i in range(m)]
        t0 = time.perf_counter()
        if opq:
            to.train_opq(Xs, m, h, 1, np.eye(d, dtype=np.float32), C0, fast=True)      # niter = 1 -> 2 loop iterations
            its = 2
        else:
            codes = oracle.encode_pq(Xs, np.concatenate([c.reshape(-1) for c in C0]), m, h)
            C1 = to.update_centers_fast(C0, Xs, codes, off, h)
            oracle.encode_pq(Xs, np.concatenate([c.reshape(-1) for c in C1]), m, h)
            its = 2
        dt = time.perf_counter() - t0
        ms_it = dt / its * 1e3 * (n / ns)
        cpu = {"value": round(n / (ms_it * 1e-3), 1), "unit": "vectors x iterations / s", "cores": os.cpu_count() or 1, "kind": "port",
               "ms_per_iteration_scaled": round(ms_it, 1),
               "sample": "%d iterations of oracle/train_oracle.py (numpy + the C oracle's encode, all host cores) on the first %d of the %d "
                         "vectors (%.1f s), time scaled by %g" % (its, ns, n, dt, n / ns)}
    obj = None
    if opq:
        o = np.asarray(out[3], dtype=np.float64)
        obj = {"first": float(o[0]), "last": float(o[-1]), "monotone_non_increasing": bool((np.diff(o) <= 1e-3 * o[:-1]).all())}
    line = {
        "metric": "%s training throughput (vectors x iterations / s)" % a.workload, "value": round(n / (ms_step * 1e-3), 1),
        "unit": "vectors x iterations / s", "n_gpus": 1, "steps": int(iters), "warmup": a.warmup, "ms_per_step": round(ms_step, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "SIFT1M-shape %s m=8 h=256, niter=%d (%s)" % (a.workload, niter, "src/OPQ.jl:49-139" if opq else "src/PQ.jl:68-99"),
                   "n": n, "d": d, "m": m, "h": h, "generator": "splitmix64 (rayuela.jl_amd/synth.py)", "parallelism": "single GPU"},
        "per_iteration_ms": per_iter,
        "outside_the_loop_ms": {"x_upload": round(prof["h2d_ms"] or fine["h2d_ms"], 2), "init": round(fine["init_ms"], 2),
                                "results_download": round(fine["d2h_ms"], 2), "whole_call_wall": round(wall * 1e3, 1)},
        "profiled_loop_ms_per_step": round(fine["loop_ms"] / fi, 4),
        "polar_factor": {"newton_schulz_steps_per_iteration": round(prof["ns_steps"] / iters, 2),
                         "jacobi_sweeps_per_iteration": round(prof["jacobi_sweeps"] / iters, 2),
                         "host_fallbacks": int(prof["host_polar"])} if opq else None,
        "objective": obj,
        "roofline": roof, "cpu_baseline": cpu,
    }
    print(json.dumps(line))
    sys.stdout.flush()


def main():
    a = parse()
    if a.workload in ("train_opq", "train_pq"):
        if a.gpus != 1:
            sys.stderr.write("bench.py: the training workloads run on one GPU\n")
            sys.exit(2)
        return train_bench(a)
    dev_list = [int(x) for x in a.devices.split(",") if x.strip() != ""] if a.devices else None
    if dev_list:
        if not a.inproc:
            sys.stderr.write("bench.py: --devices needs --inproc\n")
            sys.exit(2)
        a.gpus = len(dev_list)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1 and not a.inproc:
        sys.exit(self_spawn(a))
    world = int(env_world) if env_world is not None and not a.inproc else 1
    if not a.inproc and world != a.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; start it as `python bench.py --gpus N` or with "
                         "torch.distributed.run --nproc-per-node N\n" % (a.gpus, world))
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0")) if world > 1 else 0
    local = int(os.environ.get("LOCAL_RANK", "0")) if world > 1 else 0

    import numpy as np
    import torch
    import torch.distributed as dist
    need = (max(dev_list) + 1 if dev_list else a.gpus) if a.inproc else local + 1
    if torch.cuda.device_count() < need and os.environ.get("RQ_BENCH_BACKEND") != "gloo":
        sys.stderr.write("bench.py: %d GPUs requested, %d visible\n" % (a.gpus, torch.cuda.device_count()))
        sys.exit(2)
    # RQ_BENCH_BACKEND=gloo: debugging aid for a one-GPU box -- the ranks share the visible GPUs and the
    # collectives run on host copies, so the N > 1 control flow can be exercised without N devices
    debug_gloo = os.environ.get("RQ_BENCH_BACKEND") == "gloo"
    if debug_gloo:
        local %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if debug_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        # the ranks > 0 wait for rank 0's post-processing (checks, CPU baseline, the in-library leg over ALL devices) on the host:
        # an RCCL barrier would park a spinning kernel on every GPU that leg is about to time
        host_group = dist.new_group(backend="gloo")

    def barrier():
        if world > 1:
            dist.barrier()

    import rayuela_jl_amd as rq
    import rayuela_jl_amd.synth as synth
    import rayuela_jl_amd.synth_torch as st
    from rayuela_jl_amd import device as rqd
    from rayuela_jl_amd.sharded import ShardedIndex, shard_bounds

    ngpu = a.gpus
    wl = a.workload if a.workload != "auto" else ("pq" if ngpu == 1 else "sift1b")
    big = wl == "sift1b"
    if wl == "deep":
        d, m, name = 96, 16, "Deep1M-shape OPQ d=96 m=16 h=256"
    elif big:
        d, m, name = 128, 8, "SIFT1B-shape base (synthetic uint8 codes) m=8 h=256"
    else:
        d, m = 128, 8
        name = "SIFT1M-shape %s m=8 h=256" % ("OPQ" if wl == "opq" else "PQ")
    use_R = wl in ("opq", "deep")
    h = 256
    n = a.n or (1_000_000_000 if big else 1_000_000)
    nq = a.nq or (1024 if big else 10_000)
    K = a.k or (100 if big else 1000)
    nshards = ngpu
    bounds = shard_bounds(n, nshards)
    r0, r1 = (bounds[rank], bounds[rank + 1]) if not a.inproc else (0, n)
    n_local = r1 - r0

    # ---- synthetic inputs: the splitmix64 generators of synth.py, run on the device --------------
    def gen(rows, row0):
        if wl == "deep":
            return st.deep_like(rows, d, seed=synth.SEED_BASE, row0=row0, device=device)
        return st.sift_like(rows, d, seed=synth.SEED_BASE, ncentres=65536, row0=row0, device=device)
    # queries and the codebook training sample are rows of the same stream beyond any base row
    Q = gen(nq, 3_000_000_000)
    S = gen(20_000, 3_100_000_000)
    R = torch.from_numpy(synth.rotation(d)).to(device) if use_R else None
    S_train = rqd.rotate_T(R, S) if use_R else S
    C = synth.codebooks(S_train.cpu().numpy(), m, h, seed=synth.SEED_CODEBOOK, iters=5, sample=20000)   # harness side, untimed
    Ccat = torch.from_numpy(synth.cat_codebooks(C)).to(device)
    centers = torch.from_numpy(np.stack(C)).to(device)
    Qs = rqd.rotate_T(R, Q) if use_R else Q     # linscan_opq rotates the queries first (src/Linscan.jl:102)

    # ---- (b) encode: quantize_pq / quantize_opq of this rank's rows of the resident base -----------------------
    enc_ms = None
    X = None
    if big:
        codes = None if a.inproc else rqd.synth_codes(n_local, m, synth.SEED_BASE, row0=r0, device=device)
    else:
        X = torch.cat([gen(min(250_000, n_local - o), r0 + o) for o in range(0, n_local, 250_000)], 0)
        codes = torch.empty((n_local, m), dtype=torch.uint8, device=device)
        if use_R:
            enc = lambda: rqd.encode_opq(X, R, Ccat, m, h, out=codes)   # noqa: E731
        else:
            enc = lambda: rqd.encode_pq(X, Ccat, m, h, out=codes)       # noqa: E731
        enc_ms, _ = timed(enc, a.steps, a.warmup, barrier)

    # ---- (a) ADC scan + top-k over the whole base ------------------------------------------------------------------
    res = {}
    ix_lib = None
    if a.inproc:
        ix_lib = rq.Index(C, d, devices=dev_list or list(range(ngpu)))
        if big:
            ix_lib.set_codes_synth(n, synth.SEED_BASE)
        else:
            ix_lib.set_codes(codes.cpu().numpy())
        Qh = Q.cpu().numpy()
        Rh = None if R is None else R.cpu().numpy()

        def scan():
            res["r"] = ix_lib.search(Qh, K, R=Rh, id_base=0)
    elif world == 1 and big:
        # BASELINE config 5 on one GPU: the same resident, once-ordered shard object the N > 1 ranks hold (world size 1)
        ix = ShardedIndex(codes, centers, id_offset=0)

        def scan():
            res["r"] = ix.search(Qs, K)
    elif world == 1:
        out = (torch.empty((nq, K), dtype=torch.float32, device=device),
               torch.empty((nq, K), dtype=torch.int32, device=device))

        # the headline call: raw resident codes in, like linscan_pq's B -- the library orders a scratch copy of the base
        # INSIDE the call when that pays (nq >= 2048: csrc/rq_order.hip), so that time is part of every timed step
        def scan():
            res["r"] = rqd.linscan(codes, centers, Qs, K, out=out)
    else:
        ix = ShardedIndex(codes, centers, id_offset=r0, host_staging=debug_gloo)

        def scan():
            res["r"] = ix.search(Qs, K)
    scan_ms, scan_wall = timed(scan, a.steps, a.warmup, barrier)
    ix_info = ix_lib.info() if ix_lib is not None else None
    if os.environ.get("RQ_SCAN_STATS") and rank == 0:      # development aid: the scan kernel's phase / filter counters
        from rayuela_jl_amd import _lib
        sys.stderr.write("scan_stats %s\n" % json.dumps(_lib.scan_stats()))
    if a.inproc:
        scan_ms = scan_wall      # the library's own streams do the work: host wall clock is the step time

    # the scan kernel alone on this rank's shard (roofline): HIP events on the stream it is launched on.  The base is put in
    # bank-aware row order ONCE here (rq_dev_order_rows: what an index handle does at load time), so the launches below are
    # the scan kernel and nothing else -- the same kernel on the same ordered rows as inside the headline call.
    kern_ms = None
    kern_base = None
    order_info = None
    if not a.inproc:
        kl = min(K, n_local)
        kout = torch.empty((nq, kl), dtype=torch.int64, device=device)
        ks = max(2, min(a.steps, 10))
        if world == 1 and big and ix.ordered is not None:
            ordered = ix.ordered
        elif world > 1 and ix.ordered is not None:
            ordered = ix.ordered
        else:
            # the kind of base the timed call scans: a raw-pointer call orders its scratch copy only when that pays (from 2048
            # queries, below k = 8192: rq_scan_orders_in_call) -- at the reference's default k = 10000 it scans the rows as they
            # arrive, and so does this leg (round 5 timed an ordered base there: kernel_ms > ms_per_step)
            from rayuela_jl_amd import _lib as _lo
            in_call = int(_lo.lib().rq_scan_orders_in_call(n_local, nq, kl))     # 0 arrival order, 1 sorted, 2 sorted + balanced
            if in_call and os.environ.get("RQ_SCAN_ORDER", "1") != "0":
                # (a base ordered ONCE is always balanced -- the greedy pass of rq_order.hip --; a call that orders its own
                # copy balances it only from 16384 queries on: this leg orders the way the timed call does)
                rq.set_tuning("ORDER_GREEDY", 1 if in_call == 2 else 0)
                ordered = rqd.order_rows(codes)
                rq.set_tuning("ORDER_GREEDY", 1)
            else:
                ordered = codes
        if ordered is codes:
            kern_base = "arrival order (what the timed call scans at this shape: no in-call ordering)"
        elif world == 1 and not big and in_call == 1:
            kern_base = "bank-aware row order, sorted not balanced (what the timed call scans: it orders its own copy, and at this batch size without the greedy balance)"
        else:
            kern_base = "bank-aware row order, sorted and balanced (what the timed call scans: ordered inside the call, or once by the index)"
        kern_total, _ = timed(lambda: rqd.linscan(ordered, centers, Qs, kl, id_offset=r0, want_keys=True, out=kout), ks, 1, barrier)
        kern_ms = kern_total / ks
        if world == 1 and not big and ordered is not codes and not a.no_ab:
            # the three ways to run the same scan, same answer bit for bit (checked below):
            #   in_call   = the headline (`value`): arrival-order codes in, ordering inside every call
            #   prepared  = the base ordered once (index handle / rq_dev_order_rows), searches pay nothing for it
            #   arrival   = ordering switched off (round 3's path)
            o2 = (torch.empty((nq, K), dtype=torch.float32, device=device), torch.empty((nq, K), dtype=torch.int32, device=device))
            prepared = rqd.order_rows(codes)          # ordered ONCE: sorted and balanced, whatever the batch size
            prep_ms, _ = timed(lambda: rqd.linscan(prepared, centers, Qs, K, out=o2), a.steps, a.warmup, barrier)
            same_prep = bool(torch.equal(o2[0].view(torch.int32), res["r"][0].view(torch.int32)) and torch.equal(o2[1], res["r"][1]))
            ord_ms, _ = timed(lambda: rqd.order_rows(codes), 3, 1, barrier)
            rq.set_tuning("SCAN_ORDER", 0)
            arr_ms, _ = timed(lambda: rqd.linscan(codes, centers, Qs, K, out=o2), a.steps, a.warmup, barrier)
            rq.set_tuning("SCAN_ORDER", 1)
            same_arr = bool(torch.equal(o2[0].view(torch.int32), res["r"][0].view(torch.int32)) and torch.equal(o2[1], res["r"][1]))
            order_info = {
                "what": "bank-aware row order of the base (csrc/rq_order.hip): rows sorted by the top 3 bits of their leading code "
                        "bytes so that the 32 lanes of an LDS table gather hit distinct bank columns; a base that is ordered ONCE "
                        "(prepared: index handles, rq_dev_order_rows) is also balanced -- the rows of a sort bucket are dealt to its "
                        "lane groups so that the uncovered tables' columns fill evenly --, a call that orders its own copy does that "
                        "from 16384 queries on (it costs 0.11 ms per 1e6 rows); ids stay original row numbers",
                "in_call_ms_per_step": round(scan_ms / a.steps, 4),
                "prepared_ms_per_step": round(prep_ms / a.steps, 4), "prepared_value": round(nq / (prep_ms / a.steps * 1e-3), 1),
                "order_rows_ms_once": round(ord_ms / 3, 4),
                "arrival_order_ms_per_step": round(arr_ms / a.steps, 4), "arrival_order_value": round(nq / (arr_ms / a.steps * 1e-3), 1),
                "answers_identical": bool(same_prep and same_arr)}
            del o2, prepared
        del ordered

    vals = [scan_ms, enc_ms or 0.0, scan_wall, kern_ms or 0.0]
    if world > 1:
        t = torch.tensor(vals, dtype=torch.float64, device="cpu" if debug_gloo else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        vals = [float(x) for x in t.tolist()]
    scan_ms, enc_ms_max, scan_wall, kern_ms_max = vals
    ms_step = scan_ms / a.steps
    qps = nq / (ms_step * 1e-3)                       # true queries/s against the whole n-row base

    if rank != 0:
        # release this rank's shard before rank 0 drives every device of the node from ONE process (inproc leg below)
        res.clear()
        ix = codes = X = None      # noqa: F841
        torch.cuda.empty_cache()
        dist.barrier(group=host_group)
        dist.destroy_process_group()
        return

    # ---- roofline --------------------------------------------------------------------------------------------------
    roof = scan_roofline(m, n_local, nq, min(K, n_local), kern_ms_max) if kern_ms is not None else None
    if roof is not None:
        roof["kernel_base"] = kern_base
    encode = None
    if enc_ms is not None:
        enc_ms_step = enc_ms_max / a.steps
        enc_flops = 2.0 * d * h * n_local           # per GPU
        tf = enc_flops / (enc_ms_step * 1e-3) / 1e12
        sub_w = d // m
        from rayuela_jl_amd import _lib as _l
        enc_kernel = (_l.lib().rq_last_encode_kernel() or b"").decode() or "?"      # what the library actually launched
        split = enc_kernel in ("encode_pq_split_kernel", "encode_pq_filter_kernel")
        t_enc = enc_ms_step * 1e-3
        enc_roof = {"bound": "mfma", "kernel": enc_kernel,
                    "achieved": round(tf, 2), "peak": FP32_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(tf / FP32_PEAK_TFLOPS, 4), "traffic": None,
                    "hbm_GBps": round((4.0 * d + m) * n_local / t_enc / 1e9, 1),
                    "definition": "algorithmic 2*d*h flop per vector / time vs the f32 matrix peak (SURVEY.md 8d)"}
        if split:
            # The distance products run as an exact FILTER on the bf16 matrix cores (3, or 2 for sub <= 8, K = 16 MFMAs per 32
            # centroids x 32 vectors); pairs the filter cannot settle (2-3 %) get the canonical f32 evaluation in a second
            # launch.  What binds is instruction ISSUE on the SIMDs: a 32 x 32 x 16 bf16 MFMA holds the matrix pipe 32 cycles,
            # a wave64 VALU instruction its SIMD ~4, and in this kernel the two do not overlap (PMC: VALU-busy + MFMA-busy =
            # the kernel's cycles, profiles/r6_pmc_counters.md).  frac = those issue cycles / (SIMDs x clock x time) -- a
            # fraction of a real roof, never above 1; the SURVEY 8(d) yardsticks ride along.
            nmf = 2 if sub_w <= 8 else 3
            n_mfma = nmf * ((h + 31) // 32) * m * ((n_local + 31) // 32)        # wave-level MFMA instructions per launch
            bf_flops = 2.0 * 32 * 32 * 16 * n_mfma
            cnt = load_encode_counters(enc_kernel, sub_w)
            f32eq = {"achieved_TFLOPs": round(tf, 2), "peak": FP32_PEAK_TFLOPS, "frac": round(tf / FP32_PEAK_TFLOPS, 4),
                     "note": "SURVEY 8(d)'s yardstick (2*d*h flop per vector vs the f32 matrix peak); the products do not run at the "
                             "f32 rate, so this may pass 1 and is not a fraction of a roof"}
            # The line LEADS with the hardware roof the kernel is closest to -- the bf16 matrix pipe (work issued / 2.5 PF dense) or
            # HBM (algorithmic bytes / 8 TB/s), whichever fraction is larger -- and carries the issue-slot accounting (which says
            # WHY it is not closer: VALU + MFMA issue fill the SIMDs without overlapping) as `simd_issue`.
            bf_tf = bf_flops / t_enc / 1e12
            hbm_gbs = (4.0 * d + m) * n_local / t_enc / 1e9
            bf16 = {"issued_TFLOPs": round(bf_tf, 1), "peak": 2500.0, "frac": round(bf_tf / 2500.0, 4)}
            hbm_e = {"algorithmic_GBps": round(hbm_gbs, 1), "peak_GBps": HBM_PEAK_GBS, "frac": round(hbm_gbs / HBM_PEAK_GBS, 4)}
            if bf16["frac"] >= hbm_e["frac"]:
                enc_roof = {"bound": "mfma", "kernel": enc_kernel, "achieved": bf16["issued_TFLOPs"], "peak": 2500.0, "unit": "TFLOP/s",
                            "frac": bf16["frac"], "traffic": None,
                            "definition": "bf16 matrix-core work the filter issues (%d x v_mfma_f32_32x32x16_bf16 per 32 centroids x 32 "
                                          "vectors) per second vs the dense bf16 peak" % nmf}
            else:
                enc_roof = {"bound": "hbm", "kernel": enc_kernel, "achieved": hbm_e["algorithmic_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": hbm_e["frac"], "traffic": None,
                            "definition": "algorithmic bytes (4 d + m per vector: X read once, codes written) per second vs HBM peak"}
            enc_roof.update({"bf16_mfma": bf16, "hbm": hbm_e, "f32_equivalent": f32eq})
            issue_peak = NUM_CU * 4 * CLK_GHZ * 1e9
            simd = {"unit": "issue cycles/s", "peak": issue_peak,
                    "definition": "(VALU wave-instructions x 4 + MFMA instructions x 32 cycles) per second vs 4 SIMDs x %d CUs x %.1f GHz; "
                                  "instruction counts per launch from profiles/r6_encode_counters.json (PMC SQ_INSTS_VALU / SQ_INSTS_MFMA "
                                  "of the same library build), scaled to this run's rows.  Not a distance to a hardware roof: it says "
                                  "the SIMDs' issue slots are this full, i.e. the kernel is instruction-bound" % (NUM_CU, CLK_GHZ)}
            if cnt is not None:
                scale = float(n_local) / cnt["rows"]
                valu = (cnt["SQ_INSTS_VALU"] - cnt["SQ_INSTS_MFMA"]) * scale
                issue = (valu * 4.0 + cnt["SQ_INSTS_MFMA"] * scale * 32.0) / t_enc
                simd.update({"achieved": round(issue, 1), "frac": round(issue / issue_peak, 4),
                             "valu_insts_per_launch": round(valu), "mfma_insts_per_launch": round(cnt["SQ_INSTS_MFMA"] * scale),
                             "launches": cnt.get("launches", "tables + filter + exact pass")})
            else:
                simd.update({"achieved": None, "frac": None,
                             "note": "no instruction counters for this library build in profiles/"})
            enc_roof["simd_issue"] = simd
        if use_R:
            enc_roof["note"] = (enc_roof.get("note", "") + "; time includes the R'X rotation kernel (2*d*d f32-MFMA flop per vector more, "
                                "not counted in achieved)").lstrip("; ")
        encode = {"metric": "encode vectors/sec (%s)" % ("quantize_opq" if use_R else "quantize_pq"),
                  "value": round(n / (enc_ms_step * 1e-3), 1), "unit": "vectors/s", "ms_per_step": round(enc_ms_step, 4),
                  "roofline": enc_roof}

    # ---- checks on the returned answer (size-independent) ----------------------------------------------------------
    rd, ri = res["r"]
    rd_h = rd.cpu().numpy() if hasattr(rd, "cpu") else rd
    ri_h = (ri.cpu().numpy() if hasattr(ri, "cpu") else ri).view(np.uint32)
    checks = {"ascending": bool((np.diff(rd_h, axis=1) >= 0).all()), "ids_in_range": bool(ri_h.max() < n),
              "ids_unique_per_query": bool(all(len(np.unique(ri_h[q])) == K for q in range(min(nq, 64))))}
    if big:
        # regenerate the returned rows' codes from the hash (no device read) and redo their ADC sums on the host
        from oracle import oracle
        nchk = min(nq, 32)
        qh = Qs[:nchk].cpu().numpy()
        lut = np.stack([oracle.adc_lut(np.stack(C), qh[q]) for q in range(nchk)])       # [nchk][m][256], f32 like the reference
        ids64 = ri_h[:nchk].astype(np.uint64)
        e = ids64[:, :, None] * np.uint64(m) + np.arange(m, dtype=np.uint64)[None, None, :]
        cb = (synth.splitmix64(e ^ np.uint64(synth.SEED_BASE)) >> np.uint64(56)).astype(np.int64)
        acc = np.take_along_axis(lut[:, 0, :], cb[:, :, 0], axis=1)
        for kk in range(1, m):                                                          # sequential f32 sum, :85-87
            acc = acc + np.take_along_axis(lut[:, kk, :], cb[:, :, kk], axis=1)
        checks["returned_dists_recomputed_bit_exact"] = bool(np.array_equal(acc.view(np.uint32), rd_h[:nchk].view(np.uint32)))
        checks["queries_checked"] = nchk

    # ---- recall (sanity + parity: ids are bit-exact, so recall is identical by construction) ------------------------
    recall = None
    if world == 1 and not a.inproc and X is not None:
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
        rec = rq.eval_recall(arg.cpu().numpy(), ri_h[:nrec].astype(np.int64), K, verbose=False)
        recall = {"r@1": float(rec[0]), "r@10": float(rec[min(9, K - 1)]), "r@100": float(rec[min(99, K - 1)]),
                  "r@%d" % K: float(rec[K - 1]), "queries": nrec}

    # ---- CPU baseline on this box's host cores (rank 0, N = 1 only) -----------------------------------------------
    cpu = None
    if ngpu == 1 and not a.no_cpu:
        from oracle import oracle
        nb = min(n, 20_000_000)            # sift1b: the reference is timed on the first 2e7 rows of the base
        codes_h = (codes[:nb] if codes is not None else rqd.synth_codes(nb, m, synth.SEED_BASE, device=device)).cpu().numpy()
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
        same = None
        if nb == n:
            same = bool(np.array_equal(i_cpu, ri_h[:s1]) and np.array_equal(d_cpu.view(np.uint32), rd_h[:s1].view(np.uint32)))
        cpu = {"value": round(s1 / dt1 * (nb / n), 3), "unit": "queries/s", "cores": cores,
               "kind": "reference" if use_ref else "port",
               "sample": "%d of the %d queries against %s, k=%d (%.1f s)%s; deps/src/linscan_aqd.cpp built g++ -O3 -fopenmp "
                         "as in deps/build.jl:23" % (s1, nq, "the full %d-row base" % n if nb == n else
                                                     "the first %d of the %d rows" % (nb, n), K, dt1,
                                                     "" if nb == n else ", rate scaled by %g to the full base" % (nb / n)),
               "gpu_matches_cpu_bit_exact": same}
        # SURVEY.md 8d / BASELINE.md section 3: the reference scan at K in {1, 100, 1000}, on all host cores and on ONE thread
        # (bounded samples, ~1-2 s each; omp_set_num_threads of the libgomp the reference .so runs on)
        if nb == n and use_ref:
            import ctypes
            try:
                gomp = ctypes.CDLL("libgomp.so.1")
                by_k = []
                for kk in (1, 100, 1000):
                    if kk > n:
                        continue
                    for thr in (cores, 1):
                        gomp.omp_set_num_threads(int(thr))
                        sq = max(1, min(nq, 2 * thr if thr > 1 else 4))
                        t0 = time.perf_counter()
                        fn(codes_h, cen_h, Q_h[:sq], kk)
                        dtk = time.perf_counter() - t0
                        if dtk < 0.5 and sq < nq:      # too short to trust: one larger batch
                            sq = int(min(nq, max(sq + 1, sq * min(8.0, 1.0 / max(dtk, 1e-3)))))
                            t0 = time.perf_counter()
                            fn(codes_h, cen_h, Q_h[:sq], kk)
                            dtk = time.perf_counter() - t0
                        by_k.append({"k": kk, "threads": int(thr), "queries": sq, "seconds": round(dtk, 3),
                                     "value": round(sq / dtk, 2), "unit": "queries/s"})
                gomp.omp_set_num_threads(int(cores))
                cpu["scan_by_k_and_threads"] = by_k
            except Exception as e:   # noqa: BLE001 -- a reported baseline, never the measurement
                cpu["scan_by_k_and_threads"] = {"error": repr(e)[:200]}
        if X is not None:
            ne = min(n, 1_000_000)
            Xh = (rqd.rotate_T(R, X[:ne]) if use_R else X[:ne]).cpu().numpy()
            t0 = time.perf_counter()
            c_cpu = oracle.encode_pq(Xh, synth.cat_codebooks(C), m, h)
            dte = time.perf_counter() - t0
            cpu["encode"] = {"value": round(ne / dte, 1), "unit": "vectors/s", "kind": "port", "cores": oracle.num_threads(),
                             "sample": "%d vectors (%.2f s), oracle/rq_oracle.c" % (ne, dte),
                             "codes_match": bool(np.array_equal(c_cpu, codes[:ne].cpu().numpy()))}
            # the reference's own algorithm SHAPE (src/PQ.jl:37-43): per sub-space an sgemm into an h x n distance matrix
            # (OpenBLAS, all cores), the elementwise  max(sa + sb - 2 r, 0)  pass and a serial first-index argmin over
            # its columns (Distances.pairwise + Clustering.update_assignments!, both single-threaded loops in Julia)
            try:
                from scipy.linalg import blas as _blas
                ns = min(ne, 200_000)
                Xs_all = Xh[:ns]
                sub_off = synth.splitarray(d, m)
                t0 = time.perf_counter()
                shaped = np.empty((ns, m), dtype=np.uint8)
                for i in range(m):
                    Ci = np.ascontiguousarray(C[i], dtype=np.float32)
                    Xs = np.ascontiguousarray(Xs_all[:, sub_off[i]:sub_off[i + 1]])
                    r = _blas.sgemm(np.float32(1.0), Ci.T, Xs.T, trans_a=1)                     # h x ns  (mul!(r, a', b))
                    sa2 = np.einsum("ij,ij->i", Ci, Ci)
                    sb2 = np.einsum("ij,ij->i", Xs, Xs)
                    dm = sa2[:, None] + sb2[None, :]
                    dm -= 2.0 * r
                    np.maximum(dm, 0.0, out=dm)
                    shaped[:, i] = dm.argmin(axis=0)
                dts = time.perf_counter() - t0
                cpu["encode_reference_shaped"] = {
                    "value": round(ns / dts, 1), "unit": "vectors/s", "kind": "reference-shaped",
                    "sample": "%d vectors (%.2f s): OpenBLAS sgemm -> h x n dmat -> elementwise pass -> serial argmin, per "
                              "sub-space, as src/PQ.jl:37-43 does (numpy + scipy.linalg.blas)" % (ns, dts),
                    "codes_equal_to_gpu_fraction": round(float((shaped == codes[:ns].cpu().numpy()).mean()), 7)}
            except Exception as e:   # noqa: BLE001
                cpu["encode_reference_shaped"] = {"error": repr(e)[:200]}

    # ---- what a Julia ccall pays: the same calls on HOST arrays (PCIe in both directions inside the timed region) ----
    host = None
    if ngpu == 1 and X is not None and not a.no_host and not a.inproc:
        Xh, Qh = X.cpu().numpy(), Q.cpu().numpy()
        Rh = None if R is None else R.cpu().numpy()
        best_e = best_s = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            Bh = rq.quantize_pq_u8(Xh, C) if Rh is None else (rq.quantize_opq(Xh, Rh, C) - 1).astype(np.uint8)
            best_e = min(best_e, time.perf_counter() - t0)
        same_codes = bool(np.array_equal(Bh, codes.cpu().numpy()))
        res = None
        for _ in range(4):     # the first call allocates the page-locked result buffers of the library's pool
            res = None         # the previous answer goes back to the pool (or to the allocator) outside the timed call
            t0 = time.perf_counter()
            res = rq.linscan_pq(Bh, Qh, C, 8 * m, K) if Rh is None else rq.linscan_opq(Bh, Qh, C, 8 * m, Rh, K)
            best_s = min(best_s, time.perf_counter() - t0)
            ts = rq.last_timing()
        dh, ih = res
        host = {"note": "host pointers in, host pointers out (numpy arrays in; results in page-locked arrays of the library's pool), best of 3-4; `value` and `encode.value` above are the resident rates",
                "encode_ms": round(best_e * 1e3, 3), "encode_vectors_per_s": round(n / best_e, 1),
                "scan_ms": round(best_s * 1e3, 3), "scan_queries_per_s": round(nq / best_s, 1),
                "scan_split_ms": {k: round(v, 3) for k, v in ts.items()},
                "same_answer_as_resident": bool(same_codes and np.array_equal(ih - 1, ri_h) and
                                                np.array_equal(dh.view(np.uint32), rd_h.view(np.uint32)))}

    # ---- sift1b at N > 1: the same workload on ONE GPU (rank 0 alone), so the line carries its own scaling anchor ---
    ref1 = None
    if big and ngpu > 1 and not a.no_ref1 and n * m <= 64 * (1 << 30):
        try:
            if ix_lib is not None:
                ix_lib.close()
                ix_lib = None
            codes = None
            torch.cuda.empty_cache()
            whole = rqd.synth_codes(n, m, synth.SEED_BASE, row0=0, device=device)
            whole_o = rqd.order_rows(whole)          # like the shards of the N-GPU run: ordered once, outside the timed steps
            del whole
            o1 = (torch.empty((nq, K), dtype=torch.float32, device=device), torch.empty((nq, K), dtype=torch.int32, device=device))
            t_ms, _ = timed(lambda: rqd.linscan(whole_o, centers, Qs, K, out=o1), 2, 1, lambda: None)
            same = bool(np.array_equal(o1[1].cpu().numpy().view(np.uint32), ri_h) and
                        np.array_equal(o1[0].cpu().numpy().view(np.uint32), rd_h.view(np.uint32)))
            ref1 = {"n_gpus": 1, "ms_per_step": round(t_ms / 2, 3), "value": round(nq / (t_ms / 2 * 1e-3), 1),
                    "answer_identical_to_the_sharded_run": same}
            del whole_o
        except Exception as e:   # noqa: BLE001 -- an anchor, not the measurement
            ref1 = {"error": repr(e)[:200]}

    # ---- default N = 1 run: also the N > 1 lines' workload (1e9-row base, 1024 queries, k = 100) on this one GPU, so a
    # series of `bench.py --gpus 1, 2, 4, 8` runs has its single-GPU anchor in the N = 1 line (the headline `value`
    # stays BASELINE's SIFT1M-shape metric; efficiency over N is  value(N) / (N * scale_anchor_1gpu.value)) ----------
    anchor = None
    if ngpu == 1 and a.workload == "auto" and not a.inproc and not a.no_ref1 and a.n == 0 and a.k == 0 and a.nq == 0 and m == 8:
        try:
            nA, nqA, kA = 1_000_000_000, 1024, 100
            whole = rqd.synth_codes(nA, m, synth.SEED_BASE, row0=0, device=device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            whole_o = rqd.order_rows(whole)          # what the ranks of an N-GPU run do with their shards, once
            e1.record()
            torch.cuda.synchronize()
            del whole
            whole = whole_o
            qA = Qs[:nqA].contiguous()
            oA = (torch.empty((nqA, kA), dtype=torch.float32, device=device), torch.empty((nqA, kA), dtype=torch.int32, device=device))
            t_ms, _ = timed(lambda: rqd.linscan(whole, centers, qA, kA, out=oA), 2, 1, lambda: None)
            anchor = {"workload": "SIFT1B-shape base (synthetic uint8 codes) m=8 h=256 ADC linscan: what `--gpus N` (N > 1) shards",
                      "n_base_total": nA, "nq": nqA, "k": kA, "n_gpus": 1, "ms_per_step": round(t_ms / 2, 3),
                      "value": round(nqA / (t_ms / 2 * 1e-3), 1), "unit": "queries/s",
                      "base_ordered_once_ms": round(e0.elapsed_time(e1), 2)}
            del whole, whole_o, oA
            torch.cuda.empty_cache()
        except Exception as e:   # noqa: BLE001 -- an anchor, not the measurement
            anchor = {"error": repr(e)[:200]}

    # ---- N > 1: the SAME workload through the library's own multi-device index (rq_index_create_sharded: one process, one
    # shard per device, per-shard top-k gathered with grouped ncclSend/ncclRecv on an ncclCommInitAll clique, merged on device
    # 0) -- the path a Julia session takes (host code stays Julia: julia/RayuelaHIP.jl holds one handle), timed on the wall
    # clock after the torch.distributed ranks are done with the GPUs.  `value` above is the one-process-per-GPU harness.
    inproc_leg = None
    if world > 1:
        try:
            ndev = torch.cuda.device_count()
            devs = [i % ndev for i in range(world)] if debug_gloo else list(range(world))
            ixl = rq.Index(C, d, devices=devs)
            if big:
                ixl.set_codes_synth(n, synth.SEED_BASE)
            else:
                Xw = torch.cat([gen(min(250_000, n - o), o) for o in range(0, n, 250_000)], 0)
                cw = rqd.encode_opq(Xw, R, Ccat, m, h) if use_R else rqd.encode_pq(Xw, Ccat, m, h)
                ixl.set_codes(cw.cpu().numpy())
                del Xw, cw
            Qh = Q.cpu().numpy()
            Rh = None if R is None else R.cpu().numpy()
            out_l = {}

            def scan_lib():
                out_l["r"] = ixl.search(Qh, K, R=Rh, id_base=0)
            _, lib_wall = timed(scan_lib, a.steps, a.warmup, lambda: None)
            info = ixl.info()
            ld, li = out_l["r"]
            same = bool(np.array_equal(np.asarray(ld).view(np.uint32), res["r"][0].cpu().numpy().view(np.uint32)) and
                        np.array_equal(np.asarray(li).view(np.uint32), res["r"][1].cpu().numpy().view(np.uint32)))
            inproc_leg = {"what": "rq_index_create_sharded over devices %s from this one process (rank 0), same base / queries / k" % devs,
                          "ms_per_step": round(lib_wall / a.steps, 4), "value": round(nq / (lib_wall / a.steps * 1e-3), 1),
                          "unit": "queries/s", "clock": "host wall clock around K searches (host queries in, host results out)",
                          "exchange": info["exchange"], "shards": info["shards"], "devices": info["devices"],
                          "rccl_ranks": info["shards"] if info["exchange"] == "rccl" else 0,
                          "answer_identical": same}
            if not debug_gloo and inproc_leg["rccl_ranks"] != world:
                inproc_leg["error"] = "the library's exchange is %r over %d ranks, expected RCCL over %d" % (info["exchange"], inproc_leg["rccl_ranks"], world)
            ixl.close()
        except Exception as e:   # noqa: BLE001 -- a second measurement, never the headline
            inproc_leg = {"error": repr(e)[:300]}

    if a.inproc:
        par = "one process, rq_index_create_sharded over %d shard(s) on device list %s: exchange=%s" % (
            ngpu, dev_list or list(range(ngpu)), ix_info["exchange"] if ix_info else "?")
    elif world > 1:
        par = "one process per GPU (torch.distributed nccl=RCCL): rows sharded x%d, all_to_all of per-shard top-k keys + merge + gather to rank 0" % world
    else:
        par = "single GPU"
    line = {
        "metric": "ADC queries/sec (%s, exact top-%d%s)" % ("linscan_opq" if use_R else "linscan_pq", K, ", %.0e-row base" % n if big else ""),
        "value": round(qps, 1), "unit": "queries/s", "n_gpus": ngpu, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "debug_backend": "gloo (ranks share GPUs, collectives on host copies: NOT a measurement)" if debug_gloo and world > 1 else None,
        "logical_shards": ("device list %s repeats devices: several shards share one GPU -- a functional check, NOT a measurement" % dev_list)
                          if dev_list and len(set(dev_list)) < len(dev_list) else None,
        "config": {"workload": name + (" ADC linscan + RCCL top-k merge" if big else " encode + ADC linscan"),
                   "n_base_total": n, "n_base_per_gpu": (n + ngpu - 1) // ngpu, "nq": nq, "k": K, "d": d, "m": m, "h": h,
                   "generator": "splitmix64 (rayuela.jl_amd/synth.py; seeds base 1234, codebooks 99, rotation 7)",
                   "parallelism": par},
        "encode": encode,
        "roofline": roof,
        "cpu_baseline": cpu,
        "recall": recall,
        "checks": checks,
        "host_path": host,
        "row_order": order_info if (world == 1 and not big) else
                     ({"index_prepare_ms_once": round(ix.order_ms, 3)} if (not a.inproc and ix.order_ms is not None) else None),
        "same_workload_1gpu": ref1,
        "inproc": inproc_leg,
        "scale_anchor_1gpu": anchor,
        "wall_ms_per_step": round(scan_wall / a.steps, 4),
    }
    print(json.dumps(line))
    sys.stdout.flush()
    if world > 1:
        dist.barrier(group=host_group)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
