"""Row-sharded ADC search over the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm).

The reference has no distributed path; its only "scale the long axis" device is the 1e7-row chunking
with a carried top-k inside one query (deps/src/linscan_aqd.cpp:52-53,78-92).  The multi-GPU analogue:

  shard     rank r holds rows [offset_r, offset_r + n_r) of the base as resident uint8 codes
  scan      every rank scans its shard for ALL queries (queries are replicated) and keeps its local
            top-k as sorted packed keys  (ordered(dist) << 32 | global id)        -- no communication
  exchange  all_to_all: rank r receives, from every rank, the key lists of the queries it OWNS
            (queries are split in W contiguous blocks).  Per rank that is (W-1)/W * nq*k*8 bytes in and
            out over W-1 distinct xGMI links -- versus W times that for an all_gather of everything.
  merge     each rank merges W sorted lists per owned query (rq_dev_merge_topk).  Keys are totally
            ordered, so the result is bit-identical to a single-GPU scan of the whole base.
  gather    the owned results (nq/W * k * 8 bytes per rank) are gathered to rank 0, which returns the
            arrays the reference API returns.

`scan_fn` / `merge_fn` default to the HIP entry points.  They are injectable ONLY so the exchange
logic can be exercised on CPU with the gloo backend in tests/ (where the oracle plays the kernels);
the product never substitutes them.
"""
import torch
import torch.distributed as dist

KEY_MAX = -1  # 0xFFFFFFFFFFFFFFFF as int64


def _hip_scan(codes, centers, queries, k, id_offset):
    from . import device
    return device.linscan(codes, centers, queries, k, id_offset=id_offset, want_keys=True)


def _hip_merge(keys, k, id_base):
    from . import device
    return device.merge_topk(keys, k, id_base=id_base)


def shard_bounds(n_total, world):
    """Contiguous row shards, sizes differing by at most one row."""
    per, extra = divmod(n_total, world)
    bounds = [0]
    for r in range(world):
        bounds.append(bounds[-1] + per + (1 if r < extra else 0))
    return bounds


class ShardedIndex:
    def __init__(self, codes_local, centers, id_offset, group=None, scan_fn=None, merge_fn=None, host_staging=False,
                 always_exchange=False, order=True, keep_arrival_copy=False):
        # always_exchange: take the all_to_all / gather path even at world size 1 (exercises the RCCL collectives
        # on a one-GPU box; the product never sets it)
        self.always_exchange = always_exchange
        # host_staging: run the collectives on CPU copies (debug aid: a gloo group over ranks that share one GPU)
        self.host_staging = host_staging
        self._codes = codes_local         # [n_local][m] uint8, resident on this rank's device (see the `codes` property)
        self.centers = centers            # [m][256][sub] float32, replicated
        self.id_offset = int(id_offset)
        self.group = group
        self.scan_fn = scan_fn or _hip_scan
        self.merge_fn = merge_fn or _hip_merge
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # total rows of the base: k beyond it has no answer (the reference requires K <= N, linscan_aqd.cpp:91)
        self.n_total = int(codes_local.shape[0])
        self.n_local = int(codes_local.shape[0])
        # The shard is resident for many searches: put its rows in bank-aware order once (csrc/rq_order.hip: the scan's
        # table gathers then hit distinct LDS bank columns; ids stay original row numbers, the answer is unchanged).
        # HIP path only -- an injected scan_fn (CPU tests) sees the codes as given.
        self.ordered = None
        self.order_ms = None
        if scan_fn is None and codes_local.is_cuda and codes_local.shape[0] >= 65536 and order:
            from . import device
            t0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0[0].record()
            self.ordered = device.order_rows(codes_local)
            t0[1].record()
            torch.cuda.synchronize()
            self.order_ms = t0[0].elapsed_time(t0[1])
            # the ordered copy serves every search: the arrival-order rows are only kept on request (like the C index, which
            # frees them -- 1 GB of a 1.25e8 x 8 shard otherwise held twice; the caller's own reference is the caller's)
            if not keep_arrival_copy:
                self._codes = None
        if self.world > 1:
            t = torch.tensor([self.n_total], dtype=torch.int64, device="cpu" if host_staging else codes_local.device)
            dist.all_reduce(t, group=group)
            self.n_total = int(t.item())

    @property
    def codes(self):
        """The shard's rows in ARRIVAL order.  Released once the bank-aware ordered copy exists (`self.ordered`: rows permuted,
        `perm` maps position -> arrival row) unless the index was built with keep_arrival_copy=True; reading it then is an
        error rather than a silent None.  The caller's own tensor is untouched."""
        if self._codes is None:
            raise AttributeError("ShardedIndex.codes: the arrival-order rows were released when the ordered copy was built; "
                                 "pass keep_arrival_copy=True, or use .ordered (codes + perm)")
        return self._codes

    def local_keys(self, queries, k):
        """[nq][k] int64 sorted keys of this shard, padded with KEY_MAX when the shard has < k rows."""
        n_local = self.n_local
        nq = queries.shape[0]
        k_local = min(k, n_local)
        if k_local == 0:
            return torch.full((nq, k), KEY_MAX, dtype=torch.int64, device=queries.device)
        keys = self.scan_fn(self.ordered if self.ordered is not None else self._codes, self.centers, queries, k_local,
                            self.id_offset)
        if k_local < k:
            pad = torch.full((nq, k - k_local), KEY_MAX, dtype=torch.int64, device=keys.device)
            keys = torch.cat([keys, pad], dim=1)
        return keys.contiguous()

    def search_owned(self, queries, k, id_base=0):
        """Scan + exchange + merge.  Returns (q_lo, q_hi, dists, ids) for the queries this rank owns."""
        W, r = self.world, self.rank
        nq = queries.shape[0]
        if k < 1 or k > self.n_total:
            raise ValueError("k=%d must be in [1, total rows = %d]" % (k, self.n_total))
        keys = self.local_keys(queries, k)
        per = (nq + W - 1) // W
        if W == 1 and not self.always_exchange:
            d, i = self.merge_fn(keys.view(nq, 1, k), k, id_base)
            return 0, nq, d, i
        if per * W != nq:  # equal splits for all_to_all_single: pad the query axis
            pad = torch.full((per * W - nq, k), KEY_MAX, dtype=torch.int64, device=keys.device)
            keys = torch.cat([keys, pad], dim=0)
        dev = keys.device
        if self.host_staging:
            keys = keys.cpu()
        recv = torch.empty_like(keys)                      # [W][per][k]: block s = rank s's lists of MY queries
        if self.host_staging:                              # gloo has no all_to_all: W scatters do the same exchange
            for src in range(W):
                dist.scatter(recv.view(W, per, k)[src], list(keys.view(W, per, k).unbind(0)) if r == src else None,
                             src=src, group=self.group)
            recv = recv.to(dev)
        else:
            dist.all_to_all_single(recv, keys, group=self.group)
        mine = recv.view(W, per, k).permute(1, 0, 2).contiguous()   # [per][W][k]
        d, i = self.merge_fn(mine, k, id_base)
        q_lo = min(nq, r * per)
        q_hi = min(nq, (r + 1) * per)
        return q_lo, q_hi, d, i

    def search(self, queries, k, id_base=0):
        """On rank 0: (dists [nq][k] float32, ids [nq][k] int32 bit patterns of uint32); None elsewhere."""
        W = self.world
        nq = queries.shape[0]
        q_lo, q_hi, d, i = self.search_owned(queries, k, id_base)
        if W == 1 and not self.always_exchange:
            return d, i
        # rank 0 collects the owned blocks (nq/W * k * 8 bytes per rank) with a gather-to-root: W-1 point-to-
        # point transfers over W-1 DISTINCT xGMI links.  (An all_gather moves the same blocks to every rank and
        # RCCL runs it as a ring, i.e. (W-1)/W of the whole result through ONE link per rank -- 70 MB at W = 8,
        # k = 1000, more than the all_to_all above.)  Every rank takes the same collective path unconditionally:
        # an error is an error on all of them, never a divergence into different collectives.
        if self.host_staging:
            d, i = d.cpu(), i.cpu()
        gd = [torch.empty_like(d) for _ in range(W)] if self.rank == 0 else None
        gi = [torch.empty_like(i) for _ in range(W)] if self.rank == 0 else None
        dist.gather(d, gd, dst=0, group=self.group)
        dist.gather(i, gi, dst=0, group=self.group)
        if self.rank != 0:
            return None
        dev = queries.device
        return torch.cat(gd, dim=0)[:nq].contiguous().to(dev), torch.cat(gi, dim=0)[:nq].contiguous().to(dev)
