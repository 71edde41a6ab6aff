"""bench.py at N > 1, run the way the driver runs it (Next #1a).

The 8-GPU scaling run is the driver's; nothing here can measure it.  What these tests pin is that the literal command
path -- `python bench.py --gpus N ...` self-launching N ranks through torch.distributed.run, the sharded search, the
exchange + merge, rank 0's JSON line -- works end to end and returns the single-GPU answer.  On the one-GPU test box the
ranks share the device and the collectives run on host copies (RQ_BENCH_BACKEND=gloo, a debugging switch of bench.py);
`--inproc --devices 0,0,0,0` drives the in-library multi-device index (what a Julia session uses) with four logical
shards, through the RCCL send/recv gather when librccl loads."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=600):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, timeout=timeout,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def _common(line, n_gpus, rows, nq, k):
    assert line["n_gpus"] == n_gpus and line["steps"] == 2 and line["warmup"] == 1
    assert line["unit"] == "queries/s" and line["higher_is_better"] is True and line["scaling"] == "strong"
    assert line["value"] > 0 and line["ms_per_step"] > 0
    cfg = line["config"]
    assert cfg["n_base_total"] == rows and cfg["nq"] == nq and cfg["k"] == k and cfg["m"] == 8
    assert "SIFT1B-shape" in cfg["workload"]
    chk = line["checks"]
    assert chk["ascending"] and chk["ids_in_range"] and chk["ids_unique_per_query"]
    assert chk["returned_dists_recomputed_bit_exact"] and chk["queries_checked"] == min(nq, 32)


@pytest.mark.parametrize("ngpu,rows", [(2, 4_000_000), (4, 4_000_000), (8, 8_000_000)])
def test_driver_command_shape_one_process_per_gpu(ngpu, rows):
    """`python bench.py --gpus N --steps K --warmup W` (BASELINE config 5, shrunk): rc 0, one JSON line, the sharded
    answer identical to one scan of the whole base on one GPU.  N = 8 is the driver's SCALE command with eight ranks on
    the one device of the test box (1e6-row shards: above the ordering threshold, so every rank orders its shard)."""
    nq, k = 64, 100
    line = _run(["--gpus", str(ngpu), "--rows", str(rows), "--nq", str(nq), "--k", str(k), "--steps", "2", "--warmup", "1"],
                {"RQ_BENCH_BACKEND": "gloo"})
    _common(line, ngpu, rows, nq, k)
    assert line["debug_backend"] and "gloo" in line["debug_backend"]      # flagged as not-a-measurement
    assert "one process per GPU" in line["config"]["parallelism"]
    assert line["config"]["n_base_per_gpu"] == rows // ngpu
    ref = line["same_workload_1gpu"]
    assert ref and "error" not in ref, ref
    assert ref["answer_identical_to_the_sharded_run"] is True
    assert line["roofline"] and line["roofline"]["kernel"] == "adc_scan_kernel<8>"
    # the same workload once more through the library's own multi-device index (what a Julia session holds), from rank 0,
    # with one LOGICAL shard per rank on the test box's device; on a real node `rccl_ranks` must equal N
    lib = line["inproc"]
    assert lib and "error" not in lib, lib
    assert lib["shards"] == ngpu and lib["answer_identical"] is True and lib["ms_per_step"] > 0
    assert lib["exchange"] in ("peer", "rccl", "none")


def test_inproc_logical_shards_through_the_library_index():
    """`--inproc --devices 0,0,0,0`: the in-library sharded index (rq_index_create_sharded) with four logical shards."""
    rows, nq, k = 4_000_000, 64, 100
    line = _run(["--inproc", "--devices", "0,0,0,0", "--workload", "sift1b", "--rows", str(rows), "--nq", str(nq), "--k", str(k),
                 "--steps", "2", "--warmup", "1"])
    _common(line, 4, rows, nq, k)
    assert line["logical_shards"]
    assert "rq_index_create_sharded over 4 shard(s)" in line["config"]["parallelism"]
    ref = line["same_workload_1gpu"]
    assert ref and "error" not in ref, ref
    assert ref["answer_identical_to_the_sharded_run"] is True


def test_inproc_eight_logical_shards_over_the_rccl_transport():
    """`--inproc --devices 0,0,0,0,0,0,0,0` with EXCHANGE_SELFTEST: every logical shard but the first sends its top-k list to
    rank 0 (= the same device) through the library's ncclCommInitAll clique and ONE grouped ncclSend / ncclRecv -- the
    transport an 8-GPU node uses, carrying 7 lists (VERDICT r3 Next #7)."""
    rows, nq, k = 8_000_000, 64, 100
    line = _run(["--inproc", "--devices", "0,0,0,0,0,0,0,0", "--workload", "sift1b", "--rows", str(rows), "--nq", str(nq),
                 "--k", str(k), "--steps", "2", "--warmup", "1"], {"RQ_EXCHANGE_SELFTEST": "1"})
    _common(line, 8, rows, nq, k)
    assert "rq_index_create_sharded over 8 shard(s)" in line["config"]["parallelism"]
    assert "exchange=rccl" in line["config"]["parallelism"], line["config"]["parallelism"]
    ref = line["same_workload_1gpu"]
    assert ref and "error" not in ref, ref
    assert ref["answer_identical_to_the_sharded_run"] is True


def test_world_size_mismatch_is_an_error_not_a_silent_single_gpu_run():
    env = dict(os.environ)
    env.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT, env=env, timeout=300,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 2 and "WORLD_SIZE" in p.stderr


def test_default_command_shape_single_gpu_contract():
    """`python bench.py --steps K --warmup W` at N = 1 (the driver's BENCH run, shrunk): ONE JSON line carrying the contract's
    fields plus `roofline` and `cpu_baseline`, the GPU answer equal to the compiled reference (or the oracle port) bit for
    bit, codes equal to the oracle's."""
    line = _run(["--rows", "200000", "--nq", "512", "--k", "100", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["dtype"] == "f32"
    assert line["data"] == "synthetic" and line["vs_baseline"] is None and "workload" in line["config"]
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] in ("lds", "hbm", "mfma") and 0 < roof["frac"] and roof["kernel_ms"] > 0
    cpu = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] in ("reference", "port") and cpu["gpu_matches_cpu_bit_exact"] is True
    assert cpu["encode"]["codes_match"] is True
    enc = line["encode"]
    assert enc["value"] > 0 and enc["roofline"]["kernel"] == "encode_pq_filter_kernel"
    er = enc["roofline"]
    # the line leads with the HARDWARE roof the encode is closest to (bf16 matrix pipe or HBM: the larger fraction); the issue-slot
    # accounting that explains the gap rides along as `simd_issue`
    assert er["bound"] in ("mfma", "hbm") and 0 < er["frac"] <= 1.0
    assert 0 < er["bf16_mfma"]["frac"] <= 1.0 and 0 < er["hbm"]["frac"] <= 1.0 and "f32_equivalent" in er
    assert er["frac"] == max(er["bf16_mfma"]["frac"], er["hbm"]["frac"])
    assert er["simd_issue"]["frac"] is None or 0 < er["simd_issue"]["frac"] <= 1.0
    assert roof["kernel_base"]
    # replayed PMC traffic is bound to the kernel instantiation and library build that just ran: a number or an explained null
    assert roof["traffic"] is None or roof["traffic"] > 0
    assert roof["hbm"]["traffic_source"]
    assert line["checks"]["ascending"] and line["checks"]["ids_unique_per_query"]
    assert line["host_path"]["same_answer_as_resident"] is True
    assert line["recall"]["r@1"] > 0.2
