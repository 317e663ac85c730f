"""Multi-GPU sharding of the two hot paths: one process per GPU, torch.distributed (RCCL).

Neither path has a collective in the reference (there is no distributed code in BPMF at all;
its back-ends split work across GPUs inside one process).  Both shard embarrassingly
(SURVEY.md section 8e):

* matched filter -- templates are independent (the reference itself chunks them sequentially,
  BPMF/similarity_search.py:773-790): rank r owns a contiguous block of templates and the whole
  day of data.  No data-path collective; the only exchange is an all-gather of fixed-capacity
  peak records (KBs), because the CC matrix itself (17-170 GB) must never cross xGMI.
* backprojection -- sources are independent except for the per-sample max / arg-max: rank r
  owns a contiguous tile of the source grid (global ids via ``source_id_offset``) and the whole
  feature array; one all-reduce(MAX) of packed 64-bit keys (8 bytes per time sample) merges the
  per-rank maxima.  The key order is (beam ascending, source id descending) so that ties resolve
  to the lowest source id -- the same rule as a sequential scan over the full grid.

The key packing is written with plain torch integer ops so that exactly the same code runs on
HIP tensors over RCCL and on CPU tensors over gloo (tests/test_dist_gloo.py).
"""
import torch

_SIGN = -0x8000000000000000  # int64 with only the top bit set


def shard_bounds(n, world):
    """Balanced contiguous partition of range(n): list of (start, stop), len == world."""
    base, rem = divmod(int(n), int(world))
    out, start = [], 0
    for r in range(world):
        stop = start + base + (1 if r < rem else 0)
        out.append((start, stop))
        start = stop
    return out


def shard_bounds_weighted(costs, world):
    """Contiguous partition of range(len(costs)) into `world` blocks with balanced total cost: block
    r ends where the running cost first reaches (r + 1) / world of the total, and no block is empty
    while there are at least `world` items.  For the matched
    filter the cost of a template is its number of channels with non-zero weight (zero-weight
    channels are skipped by the kernel), SURVEY.md section 8e."""
    import numpy as np
    c = np.asarray(costs, dtype=np.float64)
    n = c.size
    cum = np.concatenate(([0.0], np.cumsum(c)))
    total = cum[-1]
    if total <= 0:
        return shard_bounds(n, world)
    cuts = [0]
    for r in range(1, world):
        k = int(np.searchsorted(cum, total * r / world, side="left"))
        cuts.append(min(max(k, cuts[-1]), n))
    cuts.append(n)
    if n >= world:      # no empty block while there are as many items as blocks (csrc/multi.hip: weighted_bounds)
        for r in range(1, world):
            cuts[r] = max(cuts[r], cuts[r - 1] + 1)
        for r in range(world - 1, 0, -1):
            cuts[r] = min(cuts[r], n - (world - r))
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def pack_max_keys(beam, arg):
    """(float32 beam, int32 source id) -> int64 keys, ordered by (beam asc, id desc)."""
    bits = beam.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    neg = (bits & 0x80000000) != 0
    ordered = torch.where(neg, (~bits) & 0xFFFFFFFF, bits | 0x80000000)
    low = 0xFFFFFFFF - (arg.to(torch.int64) & 0xFFFFFFFF)
    key = (ordered << 32) | low          # unsigned order, stored in int64 bit pattern
    return key ^ _SIGN                    # flip the top bit: unsigned order == signed order


def unpack_max_keys(packed):
    key = packed ^ _SIGN
    ordered = (key >> 32) & 0xFFFFFFFF
    pos = (ordered & 0x80000000) != 0
    bits = torch.where(pos, ordered & 0x7FFFFFFF, (~ordered) & 0xFFFFFFFF)
    bits32 = torch.where(bits >= 0x80000000, bits - 0x100000000, bits).to(torch.int32)
    beam = bits32.view(torch.float32)
    arg = (0xFFFFFFFF - (key & 0xFFFFFFFF)).to(torch.int32)
    return beam, arg


def allreduce_max(beam, arg, group=None):
    """Global (max beam, lowest arg-max id) across ranks; returns new tensors on every rank."""
    import torch.distributed as dist
    packed = pack_max_keys(beam, arg)
    dist.all_reduce(packed, op=dist.ReduceOp.MAX, group=group)
    return unpack_max_keys(packed)


def allgather_records(records, count, group=None):
    """All-gather of fixed-capacity record buffers.

    records: (capacity, width) tensor, the first `count` rows valid.  Returns the list of
    per-rank valid slices (every rank gets all of them).
    """
    import torch.distributed as dist
    world = dist.get_world_size(group)
    cnt = torch.tensor([int(count)], dtype=torch.int64, device=records.device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    bufs = [torch.empty_like(records) for _ in range(world)]
    dist.all_gather(bufs, records.contiguous(), group=group)
    return [b[: int(c.item())] for b, c in zip(bufs, counts)]


def allgather_varlen(records, group=None):
    """All-gather of per-rank record lists of DIFFERENT lengths: records (n_r, width) on rank r ->
    the list of every rank's records, on every rank.  The buffer is sized by an all-reduce(MAX) of the
    counts (nothing is dropped, nothing is sized by a guess); KBs on the wire -- the "RCCL all-gather of
    CC peaks" of BASELINE configs[3]."""
    import torch.distributed as dist
    n = int(records.shape[0])
    kmax = torch.tensor([n], dtype=torch.int64, device=records.device)
    dist.all_reduce(kmax, op=dist.ReduceOp.MAX, group=group)
    cap = max(1, int(kmax.item()))
    buf = torch.zeros((cap,) + tuple(records.shape[1:]), dtype=records.dtype, device=records.device)
    if n:
        buf[:n] = records
    return allgather_records(buf, n, group=group)


def broadcast_day(array, src, device, group=None):
    """The day of data / features, replicated ONCE over the group's fabric (RCCL over xGMI; gloo in the CPU
    tests): rank `src` passes the (S, C, N) float32 array (NumPy or tensor, host or device) and uploads
    it; every other rank passes None, allocates the same shape on `device` and receives it -- its host
    never reads the day, where N ranks each pulling 2-4 GB through pageable host memory would contend for
    the same few CPU cores ("broadcast once per day").  Returns the tensor on `device`."""
    import numpy as np
    import torch.distributed as dist
    rank = dist.get_rank(group)
    gsrc = src if group is None else dist.get_global_rank(group, src)
    shape = torch.zeros(4, dtype=torch.int64, device=device)
    t = None
    problem = None
    if rank == src:
        # (a bad argument on the source must not leave the other ranks inside the shape broadcast until the backend's
        # timeout: the source broadcasts a status word -- shape[3] = 0 -- and EVERY rank raises behind it)
        if array is None:
            problem = f"rank {src} is the source of the broadcast and must pass the array"
        else:
            t = array if isinstance(array, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(array, dtype=np.float32))
            t = t.to(device=device, dtype=torch.float32).contiguous()
            if t.dim() != 3:
                problem = "the day must be (S, C, N)"
            else:
                shape[:3] = torch.tensor(t.shape, dtype=torch.int64)
                shape[3] = 1
    dist.broadcast(shape, src=gsrc, group=group)
    if int(shape[3].item()) != 1:
        raise ValueError(problem or f"broadcast_day: rank {src} (the source) reported a bad argument; nothing was broadcast")
    if rank != src:
        t = torch.empty(tuple(int(x) for x in shape[:3].tolist()), dtype=torch.float32, device=device)
    dist.broadcast(t, src=gsrc, group=group)
    return t


class ShardedBeamformer:
    """Backprojection with the source grid tiled across the ranks of a process group."""

    def __init__(self, moveouts, weights_sources, group=None, device=None, local_factory=None):
        """`local_factory(moveouts_block, weights_block, source_id_offset)`: the per-rank engine (run /
        close like BeamformerGPU); None = a BeamformerGPU on `device`.  (The CPU tests of the exchange
        pass an oracle-backed stand-in.)"""
        import torch.distributed as dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.K = moveouts.shape[0]
        self.k0, self.k1 = shard_bounds(self.K, self.world)[self.rank]
        if local_factory is None:
            from .beampower import BeamformerGPU
            self.local = BeamformerGPU(moveouts[self.k0:self.k1], weights_sources[self.k0:self.k1],
                                       device=device, source_id_offset=self.k0)
        else:
            self.local = local_factory(moveouts[self.k0:self.k1], weights_sources[self.k0:self.k1], self.k0)

    def broadcast_features(self, features, src=0):
        """The day of features from rank `src` to every rank (broadcast_day); returns this rank's tensor."""
        dev = getattr(self.local, "device", torch.device("cpu"))
        return broadcast_day(features, src, dev, self.group)

    def run(self, features, weights_phases, reduce="max", out_of_bounds="strict"):
        if reduce == "none":
            # no exchange: every rank returns its own (K_local, N) slab
            return self.local.run(features, weights_phases, "none", out_of_bounds)
        beam, arg = self.local.run(features, weights_phases, "max", out_of_bounds)
        if self.world == 1:
            return beam, arg
        return allreduce_max(beam, arg, self.group)

    def close(self):
        self.local.close()


class ShardedMatchedFilter:
    """Matched filter with the templates block-partitioned across the ranks of a group."""

    def __init__(self, group=None, device=None, local=None):
        """`local`: the per-rank engine (set_data / run like MatchedFilterGPU); None = a
        MatchedFilterGPU on `device`.  (The CPU tests of the exchange logic pass a stand-in.)"""
        import torch.distributed as dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        if local is None:
            from .matched_filter import MatchedFilterGPU
            local = MatchedFilterGPU(device=device)
        self.local = local

    def set_data(self, data):
        self.local.set_data(data)

    def record_device(self):
        """Where this rank's records live for the all-gather (the engine's GPU over RCCL, the host over gloo)."""
        return getattr(self.local, "device", torch.device("cpu"))

    def set_data_broadcast(self, data, src=0):
        """`data` on rank `src` only (None elsewhere): uploaded there, broadcast over the group, adopted
        by every rank's engine without a host copy (broadcast_day)."""
        self.local.set_data(broadcast_day(data, src, self.record_device(), self.group))

    def template_range(self, n_templates, weights=None):
        """This rank's block of templates: equal counts, or -- given the (T, S, C) weights -- equal
        numbers of weighted channels (the kernel skips zero-weight channels)."""
        if weights is not None:
            import numpy as np
            w = weights.detach().cpu().numpy() if hasattr(weights, "detach") else np.asarray(weights)
            costs = (w.reshape(w.shape[0], -1) != 0).sum(axis=1)
            return shard_bounds_weighted(costs, self.world)[self.rank]
        return shard_bounds(n_templates, self.world)[self.rank]

    def run(self, templates, moveouts, weights, step=1, network_sum=True, balance=True):
        """CC of this rank's block of templates; returns (t0, t1, cc_local)."""
        t0, t1 = self.template_range(templates.shape[0], weights if balance else None)
        if t1 == t0:
            return t0, t1, None
        cc = self.local.run(templates[t0:t1], moveouts[t0:t1], weights[t0:t1], step, network_sum)
        return t0, t1, cc
