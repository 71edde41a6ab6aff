"""GPU parity tests of the multi-device index behind the C ABI (rq_index_*): a base cut into several
shards -- here LOGICAL shards of the one GPU the box has, the same code path as distinct devices up to the
transport -- must return exactly what one scan of the whole base returns: ids and distance bits."""
import os

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def _eq_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


@pytest.mark.parametrize("nshards", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("name", ["scan_sift_mini", "scan_deep_mini", "scan_dups", "scan_all_ties"])
def test_logical_shards_equal_reference_golden(rq, name, nshards):
    g = golden(name)
    m = g["codes"].shape[1]
    d = g["queries"].shape[1]
    C = [g["centers"][i] for i in range(m)]
    with rq.Index(C, d, devices=[0] * nshards) as ix:
        ix.set_codes(g["codes"])
        info = ix.info()
        assert info["shards"] == nshards and info["devices"] == 1 and sum(info["rows_per_shard"]) == g["codes"].shape[0]
        for K in g["Ks"]:
            dists, ids = ix.search(g["queries"], int(K), id_base=0)
            assert np.array_equal(ids, g["ids_K%d" % K]), (name, nshards, K)
            assert _eq_bits(dists, g["dists_K%d" % K]), (name, nshards, K)


def test_shards_shorter_than_k_and_empty_shards(rq, oracle):
    """n = 37 rows over 8 shards (4-5 rows each) with k = 20 > every shard, and n = 5 over 8 shards (3 empty)."""
    import rayuela_jl_amd.synth as synth
    m, d = 8, 32
    X = synth.sift_like(2000, d, seed=5)
    C = synth.codebooks(X, m, 256, seed=6, iters=1, sample=2000)
    Q = synth.sift_like(9, d, seed=7)
    for n, k in [(37, 20), (5, 5), (300, 64)]:
        codes = synth.random_codes(n, m, seed=11 + n)
        d0, i0 = oracle.linscan_aqd_query(codes, np.stack(C), Q, k)
        with rq.Index(C, d, devices=[0] * 8) as ix:
            ix.set_codes(codes)
            dists, ids = ix.search(Q, k, id_base=0)
        assert np.array_equal(ids, i0), (n, k)
        assert _eq_bits(dists, d0), (n, k)


@pytest.mark.parametrize("nshards", [3, 4])
def test_sharded_equals_single_scan_large_k_and_offsets(rq, nshards):
    """2e5 random rows, k = 1000 and k = 3000 (sample-sort finish + big merge), id_offset, one-based ids, OPQ."""
    import rayuela_jl_amd.synth as synth
    m, d, n, nq = 8, 64, 200_000, 40
    X = synth.sift_like(4000, d, seed=21)
    C = synth.codebooks(X, m, 256, seed=22, iters=1, sample=4000)
    Q = synth.sift_like(nq, d, seed=23)
    R = synth.rotation(d, seed=3)
    codes = synth.random_codes(n, m, seed=99)
    with rq.Index(C, d) as one, rq.Index(C, d, devices=[0] * nshards) as many:
        one.set_codes(codes, id_offset=1000)
        many.set_codes(codes, id_offset=1000)
        for k in (1000, 3000):
            d1, i1 = one.search(Q, k)
            d2, i2 = many.search(Q, k)
            assert np.array_equal(i1, i2) and _eq_bits(d1, d2), (nshards, k)
            assert i1.min() >= 1001
        d1, i1 = one.search(Q, 100, R=R)
        d2, i2 = many.search(Q, 100, R=R)
        assert np.array_equal(i1, i2) and _eq_bits(d1, d2)
    # the handle's single-shard answer is the host-pointer API's answer
    d3, i3 = rq.linscan_pq(codes, Q, C, 8 * m, 1000)
    with rq.Index(C, d) as one:
        one.set_codes(codes)
        d1, i1 = one.search(Q, 1000)
    assert np.array_equal(i1, i3) and _eq_bits(d1, d3)


def test_synth_base_on_device_matches_host_generator(rq, oracle):
    import rayuela_jl_amd.synth as synth
    m, d, n, nq, k = 8, 128, 50_000, 16, 100
    X = synth.sift_like(4000, d, seed=31)
    C = synth.codebooks(X, m, 256, seed=32, iters=1, sample=4000)
    Q = synth.sift_like(nq, d, seed=33)
    codes = synth.random_codes(n, m, seed=synth.SEED_BASE)
    d0, i0 = oracle.linscan_aqd_query(codes, np.stack(C), Q, k)
    with rq.Index(C, d, devices=[0, 0, 0]) as ix:
        ix.set_codes_synth(n, synth.SEED_BASE)
        dists, ids = ix.search(Q, k, id_base=0)
    assert np.array_equal(ids, i0) and _eq_bits(dists, d0)


def test_env_device_list_shards_the_stock_entry_points(rq, monkeypatch):
    """RAYUELA_HIP_DEVICES with more than one entry: linscan_pq / linscan_opq / the legacy symbol run the sharded path."""
    g = golden("scan_sift_mini")
    m = g["codes"].shape[1]
    C = [g["centers"][i] for i in range(m)]
    monkeypatch.setenv("RAYUELA_HIP_DEVICES", "0,0,0")
    for K in g["Ks"]:
        dists, ids = rq.linscan_aqd_query(g["codes"], g["centers"], g["queries"], int(K))
        assert np.array_equal(ids, g["ids_K%d" % K]) and _eq_bits(dists, g["dists_K%d" % K])
    dists, idx = rq.linscan_pq(g["codes"], g["queries"], C, 8 * m, 100)
    assert np.array_equal(idx, g["ids_K100"] + 1) and _eq_bits(dists, g["dists_K100"])
    # the sharded index behind these calls is kept between calls: other codebooks, another base size, another device
    # list and a release in between must all give the single-device answers
    from rayuela_jl_amd import _lib
    C2 = [c * np.float32(0.5) + np.float32(1.0) for c in C]
    half = g["codes"][: g["codes"].shape[0] // 2]
    monkeypatch.delenv("RAYUELA_HIP_DEVICES")
    d_ref, i_ref = rq.linscan_pq(half, g["queries"], C2, 8 * m, 100)
    for devs in ("0,0,0", "0,0,0", "0,0", "0,0,0,0,0"):
        monkeypatch.setenv("RAYUELA_HIP_DEVICES", devs)
        d2, i2 = rq.linscan_pq(half, g["queries"], C2, 8 * m, 100)
        assert np.array_equal(i2, i_ref) and _eq_bits(d2, d_ref), devs
        if devs == "0,0":
            assert _lib.lib().rq_release_workspaces() == 0
    dists, idx = rq.linscan_pq(g["codes"], g["queries"], C, 8 * m, 100)
    assert np.array_equal(idx, g["ids_K100"] + 1) and _eq_bits(dists, g["dists_K100"])


def test_index_argument_errors(rq):
    g = golden("scan_sift_mini")
    m = g["codes"].shape[1]
    d = g["queries"].shape[1]
    C = [g["centers"][i] for i in range(m)]
    with pytest.raises(rq.RayuelaHipError):
        rq.Index(C, d, devices=[0, 99])
    with rq.Index(C, d, devices=[0, 0]) as ix:
        with pytest.raises(rq.RayuelaHipError):
            ix.search(g["queries"], 10)                       # no codes yet
        ix.set_codes(g["codes"])
        with pytest.raises(rq.RayuelaHipError):
            ix.search(g["queries"], g["codes"].shape[0] + 1)  # k > n
        with pytest.raises(rq.RayuelaHipError):
            ix.search(g["queries"], 0)
        # row ids must stay below 0xFFFFFFFF (the padding key's id): set_codes and search use the SAME bound, so a base
        # that loads can always be searched (id_offset + n == 2^32 used to load and then fail every search)
        n = g["codes"].shape[0]
        with pytest.raises(rq.RayuelaHipError):
            ix.set_codes(g["codes"], id_offset=2 ** 32 - n)
        ix.set_codes(g["codes"], id_offset=2 ** 32 - 1 - n)
        dists, ids = ix.search(g["queries"], 10, id_base=0)
        assert np.array_equal(ids, g["ids_K10"] + np.uint32(2 ** 32 - 1 - n)) and _eq_bits(dists, g["dists_K10"])


def test_env_single_device_restores_the_callers_device(rq, monkeypatch):
    """RAYUELA_HIP_DEVICES with ONE entry switched the calling thread's device and left it switched."""
    import torch
    g = golden("scan_sift_mini")
    m = g["codes"].shape[1]
    C = [g["centers"][i] for i in range(m)]
    monkeypatch.setenv("RAYUELA_HIP_DEVICES", "0")
    before = torch.cuda.current_device()
    dists, idx = rq.linscan_pq(g["codes"], g["queries"], C, 8 * m, 100)
    assert np.array_equal(idx, g["ids_K100"] + 1) and _eq_bits(dists, g["dists_K100"])
    assert torch.cuda.current_device() == before


def test_two_host_threads_two_streams_do_not_interfere(rq):
    """scratch is keyed by (device, stream) and launches are serialised per device."""
    import threading
    g = golden("scan_sift_mini")
    m = g["codes"].shape[1]
    d = g["queries"].shape[1]
    C = [g["centers"][i] for i in range(m)]
    errs = []

    def worker(seed):
        try:
            with rq.Index(C, d, devices=[0, 0]) as ix:
                ix.set_codes(g["codes"])
                for _ in range(20):
                    for K in (10, 1000):
                        dists, ids = ix.search(g["queries"], K, id_base=0)
                        if not (np.array_equal(ids, g["ids_K%d" % K]) and _eq_bits(dists, g["dists_K%d" % K])):
                            errs.append((seed, K))
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[:3]


def test_rccl_transport_selftest_on_one_gpu(rq):
    """The RCCL exchange itself (dlopen of librccl, single-process communicator, grouped ncclSend/ncclRecv of uint64
    key lists) cannot meet a second GPU on this box; with EXCHANGE_SELFTEST every logical shard but the first sends
    its list to rank 0 = itself through that very code path."""
    g = golden("scan_sift_mini")
    m = g["codes"].shape[1]
    d = g["queries"].shape[1]
    C = [g["centers"][i] for i in range(m)]
    rq.set_tuning("EXCHANGE_SELFTEST", 1)
    try:
        with rq.Index(C, d, devices=[0, 0, 0, 0]) as ix:
            ix.set_codes(g["codes"])
            info = ix.info()
            if info["exchange"] != "rccl":
                pytest.skip("librccl could not be loaded / initialised on this box")
            for K in g["Ks"]:
                dists, ids = ix.search(g["queries"], int(K), id_base=0)
                assert np.array_equal(ids, g["ids_K%d" % K]) and _eq_bits(dists, g["dists_K%d" % K]), K
    finally:
        rq.set_tuning("EXCHANGE_SELFTEST", 0)


@pytest.mark.parametrize("selftest", [0, 1])
@pytest.mark.parametrize("chunks", [2, 3, 8])
def test_query_chunk_pipeline(rq, chunks, selftest):
    """Long top-k lists are exchanged in query chunks (scan of chunk c+1 | transfer of chunk c | merge of chunk c-1 on
    three streams); IDX_QCHUNKS forces the chunk count, with EXCHANGE_SELFTEST the chunks travel through RCCL."""
    g = golden("scan_sift_mini")
    m = g["codes"].shape[1]
    d = g["queries"].shape[1]
    C = [g["centers"][i] for i in range(m)]
    rq.set_tuning("IDX_QCHUNKS", chunks)
    rq.set_tuning("EXCHANGE_SELFTEST", selftest)
    try:
        with rq.Index(C, d, devices=[0, 0, 0]) as ix:
            ix.set_codes(g["codes"])
            if selftest and ix.info()["exchange"] != "rccl":
                pytest.skip("librccl could not be loaded / initialised on this box")
            for rep in range(2):
                for K in g["Ks"]:
                    dists, ids = ix.search(g["queries"], int(K), id_base=0)
                    assert np.array_equal(ids, g["ids_K%d" % K]) and _eq_bits(dists, g["dists_K%d" % K]), K
        # a shard shorter than k (KEY_MAX padding) inside a chunked search
        n_small = 40
        with rq.Index(C, d, devices=[0] * 7) as ix:
            ix.set_codes(g["codes"][:n_small])
            dists, ids = ix.search(g["queries"], 20, id_base=0)
            d1, i1 = rq.linscan_aqd_query(g["codes"][:n_small], g["centers"], g["queries"], 20)
            assert np.array_equal(ids, i1) and _eq_bits(dists, d1)
    finally:
        rq.set_tuning("IDX_QCHUNKS", 0)
        rq.set_tuning("EXCHANGE_SELFTEST", 0)


def test_torch_distributed_rccl_collectives_at_world_size_one(rq):
    """bench.py --gpus N runs rayuela.jl_amd/sharded.py over torch.distributed (backend nccl == RCCL).  A one-GPU box
    cannot host two ranks (RCCL refuses a duplicate GPU), but the very collectives of the N-rank path --
    all_reduce of the row count, all_to_all_single of int64 key lists, gather to rank 0 -- run here in a group of one."""
    import socket
    import torch
    import torch.distributed as dist
    from rayuela_jl_amd.sharded import ShardedIndex
    from rayuela_jl_amd import device as rqd
    g = golden("scan_sift_mini")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        codes = torch.from_numpy(g["codes"]).cuda()
        centers = torch.from_numpy(g["centers"]).cuda()
        Q = torch.from_numpy(g["queries"]).cuda()
        ix = ShardedIndex(codes, centers, id_offset=0, always_exchange=True)
        assert ix.n_total == g["codes"].shape[0]
        for K in g["Ks"]:
            d, i = ix.search(Q, int(K))
            assert np.array_equal(i.cpu().numpy().view(np.uint32), g["ids_K%d" % K]), K
            assert _eq_bits(d.cpu().numpy(), g["dists_K%d" % K]), K
    finally:
        dist.destroy_process_group()


def test_row_ids_beyond_2_to_31_with_logical_shards(rq, oracle):
    """2.2e9 rows (17.6 GB of synthetic codes) on ONE device as two logical shards: per-shard row counts stay below
    2^31 (the kernels' limit), global ids are uint32 up to 4.29e9.  Size-independent check: every returned distance
    is recomputed from the hash-generated code of the returned id; the lists must be ascending and cross 2^31."""
    import torch
    import rayuela_jl_amd.synth as synth
    if torch.cuda.mem_get_info(0)[0] < 30 * (1 << 30):
        pytest.skip("needs 30 GB of free device memory")
    m, d, n, nq, k = 8, 32, 2_200_000_000, 8, 64
    X = synth.sift_like(4000, d, seed=41)
    C = synth.codebooks(X, m, 256, seed=42, iters=1, sample=4000)
    Q = synth.sift_like(nq, d, seed=43)
    with rq.Index(C, d, devices=[0, 0]) as ix:
        ix.set_codes_synth(n, 777)
        assert ix.info()["rows_per_shard"] == [n // 2, n // 2]
        dists, ids = ix.search(Q, k, id_base=0)
    assert (np.diff(dists, axis=1) >= 0).all() and ids.dtype == np.uint32
    assert (ids >= 2 ** 31).any() and (ids < 2 ** 31).any() and int(ids.max()) < n
    lut = np.stack([oracle.adc_lut(np.stack(C), Q[q]) for q in range(nq)])
    e = ids.astype(np.uint64)[:, :, None] * np.uint64(m) + np.arange(m, dtype=np.uint64)[None, None, :]
    cb = (synth.splitmix64(e ^ np.uint64(777)) >> np.uint64(56)).astype(np.int64)
    acc = np.take_along_axis(lut[:, 0, :], cb[:, :, 0], axis=1)
    for kk in range(1, m):
        acc = acc + np.take_along_axis(lut[:, kk, :], cb[:, :, kk], axis=1)
    assert _eq_bits(acc, dists)
