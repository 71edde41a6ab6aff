"""CPU, world_size 2 and 3, gloo: the row-sharded search (shard -> local top-k keys -> all_to_all ->
merge -> gather) gives the single-scan answer bit for bit.  The kernels are played by the oracle
through ShardedIndex's injection points (tests only); what is under test is the N>1 host logic."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _f2ord(d):
    u = d.view(np.uint32).astype(np.uint64)
    return np.where(u >> 31 != 0, u ^ np.uint64(0xFFFFFFFF), u ^ np.uint64(0x80000000))


def _ord2f(o):
    o = o.astype(np.uint64)
    u = np.where(o & np.uint64(0x80000000) != 0, o ^ np.uint64(0x80000000), (~o) & np.uint64(0xFFFFFFFF))
    return u.astype(np.uint32).view(np.float32)


def _worker(rank, world, port, n, nq, K, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    import rayuela_jl_amd.synth as synth
    from rayuela_jl_amd.sharded import ShardedIndex, shard_bounds

    m, sub = 8, 4
    rng = np.random.default_rng(1)
    centers = rng.standard_normal((m, 256, sub)).astype(np.float32)
    queries = rng.standard_normal((nq, m * sub)).astype(np.float32)
    codes = synth.random_codes(n, m, seed=5)
    codes[n // 3:n // 3 + 50] = codes[0]           # duplicates -> distance ties across shards

    def scan_fn(codes_t, centers_t, queries_t, k, id_offset):   # oracle stands in for rq_dev_linscan
        d, i = oracle.linscan_aqd_query(codes_t.numpy(), centers_t.numpy(), queries_t.numpy(), k)
        keys = (_f2ord(d) << np.uint64(32)) | (i.astype(np.uint64) + np.uint64(id_offset))
        return torch.from_numpy(keys.view(np.int64).copy())

    def merge_fn(keys_t, k, id_base):                           # ... and for rq_dev_merge_topk
        ku = np.sort(keys_t.numpy().view(np.uint64).reshape(keys_t.shape[0], -1), axis=1)[:, :k]
        d = _ord2f(ku >> np.uint64(32))
        i = (ku & np.uint64(0xFFFFFFFF)).astype(np.uint32) + np.uint32(id_base)
        return torch.from_numpy(d.copy()), torch.from_numpy(i.view(np.int32).copy())

    b = shard_bounds(n, world)
    ix = ShardedIndex(torch.from_numpy(codes[b[rank]:b[rank + 1]].copy()), torch.from_numpy(centers),
                      b[rank], scan_fn=scan_fn, merge_fn=merge_fn)
    res = ix.search(torch.from_numpy(queries), K, id_base=1)
    if rank == 0:
        d0, i0 = oracle.linscan_aqd_query(codes, centers, queries, K)
        ok = np.array_equal(res[1].numpy().view(np.uint32), i0 + 1) and \
            np.array_equal(res[0].numpy().view(np.uint32), d0.view(np.uint32))
        with open(out_path, "w") as f:
            f.write("OK" if ok else "MISMATCH")
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,nq,K", [(2, 5000, 11, 100), (3, 2000, 7, 900), (2, 64, 4, 50), (8, 3000, 13, 200)])
def test_sharded_search_matches_single_scan(tmp_path, world, n, nq, K):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(world, _free_port(), n, nq, K, out), nprocs=world, join=True)
    assert open(out).read() == "OK"


def _worker_k_too_large(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rayuela_jl_amd.sharded import ShardedIndex
    ix = ShardedIndex(torch.zeros((10, 8), dtype=torch.uint8), torch.zeros((8, 256, 4)), 10 * rank,
                      scan_fn=lambda *a: None, merge_fn=lambda *a: None)
    try:
        ix.search(torch.zeros((3, 32)), 21)          # 20 rows in total
        verdict = "NO ERROR"
    except ValueError:
        verdict = "OK"
    if rank == 0:
        open(out_path, "w").write(verdict if ix.n_total == 20 else "BAD TOTAL")
    dist.barrier()
    dist.destroy_process_group()


def test_k_beyond_the_total_row_count_raises_on_every_rank(tmp_path):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker_k_too_large, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "OK"


def test_shard_bounds():
    from rayuela_jl_amd.sharded import shard_bounds
    assert shard_bounds(10, 3) == [0, 4, 7, 10]
    assert shard_bounds(8, 8) == list(range(9))
    assert shard_bounds(3, 4) == [0, 1, 2, 3, 3]
