"""Data parallelism for the MKGC hot path: one process per GPU, batch (triples) sharded, one
all-reduce of the flat gradient buffer per step over RCCL/xGMI (SURVEY.md par. 8e).

The reference has no working distributed path (src/distributed_utils.py is imported by nothing);
the only behaviour to match is "G-GPU loss/gradients equal the 1-GPU run on the same global
batch", which holds because the margin loss is a SUM over triples (model/MSNEA.py:122): summing
per-rank gradients of per-rank partial sums needs no division.
"""
import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of n_items for `rank`; the first n_items % world ranks get one extra item."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(tensors, rank: int, world: int):
    """Slice every [B, ...] tensor of a batch to this rank's triples."""
    B = tensors[0].shape[0]
    lo, hi = shard_range(B, rank, world)
    return tuple(t[lo:hi].contiguous() for t in tensors)


class RowShard:
    """Ownership of the N entity rows by `world` ranks in equal padded shards: rank g owns rows [g*n_shard, g*n_shard + n_loc).

    SURVEY.md par. 8(e): "row-sharded N/G + all-gather of [N,D] read-outs".  Every rank projects and fuses ITS rows only, the
    compact [n_shard, D] read-outs are all-gathered into one [n_pad, D] table the scorer reads, and the scorer's [n_pad, D]
    gradient is reduce-scattered back to the owners.  Shards are padded to a common size (the collectives need equal chunks);
    pad rows (>= N) are never indexed by a triple."""

    def __init__(self, n_rows: int, rank: int = 0, world: int = 1, group=None, force: bool = False):
        if world < 1 or not (0 <= rank < world):
            raise ValueError(f"bad rank/world {rank}/{world}")
        self.n_rows, self.rank, self.world, self.group = int(n_rows), int(rank), int(world), group
        self.force = bool(force)       # run the collectives even with one rank (exercises the RCCL path on a 1-GPU box)
        self.n_shard = -(-self.n_rows // self.world)
        self.n_pad = self.n_shard * self.world
        self.lo = self.rank * self.n_shard
        self.n_loc = max(0, min(self.n_rows, self.lo + self.n_shard) - self.lo)

    @property
    def sharded(self):
        return self.world > 1 or self.force

    def all_gather(self, full: torch.Tensor, local: torch.Tensor):
        """full [n_pad, D] <- every rank's local [n_shard, D] (stream-ordered on the GPU backends: the compute stream waits for
        the collective, the host does not)."""
        if tuple(full.shape[:1]) != (self.n_pad,) or local.shape[0] != self.n_shard:
            raise ValueError(f"all_gather: full {tuple(full.shape)} / local {tuple(local.shape)} do not match shards of {self.n_shard}")
        if full.is_cuda and dist.get_backend(self.group) == "gloo":      # test transport only (two ranks sharing one GPU): gloo moves host buffers
            host = torch.empty(full.shape, dtype=full.dtype)
            dist.all_gather_into_tensor(host, local.cpu(), group=self.group)
            full.copy_(host)
            return
        dist.all_gather_into_tensor(full, local, group=self.group)

    def reduce_scatter(self, local: torch.Tensor, full: torch.Tensor):
        """local [n_shard, D] <- sum over ranks of their full [n_pad, D], rows of this rank's shard."""
        if tuple(full.shape[:1]) != (self.n_pad,) or local.shape[0] != self.n_shard:
            raise ValueError(f"reduce_scatter: full {tuple(full.shape)} / local {tuple(local.shape)} do not match shards of {self.n_shard}")
        if full.is_cuda and dist.get_backend(self.group) == "gloo":      # test transport only, as above
            host = torch.empty(local.shape, dtype=local.dtype)
            dist.reduce_scatter_tensor(host, full.cpu(), op=dist.ReduceOp.SUM, group=self.group)
            local.copy_(host)
            return
        dist.reduce_scatter_tensor(local, full, op=dist.ReduceOp.SUM, group=self.group)


def common_step_count(n_local: int, batch_size: int, group=None) -> int:
    """Number of optimiser steps EVERY rank runs this epoch: ceil(max over ranks of n_local / batch_size).

    Shards handed out by ``shard_range`` differ by one item, so ``range(0, n_local, batch_size)`` can differ by one step between
    ranks (1025 vs 1024 items at batch 1024): the rank with the extra step would then wait for ever in a gradient all-reduce
    nobody else enters.  One tiny MAX all-reduce per epoch settles the count."""
    if batch_size < 1:
        raise ValueError(f"batch_size={batch_size}")
    steps = -(-max(0, int(n_local)) // batch_size)
    if dist.is_available() and dist.is_initialized():       # also with one rank: the same collective on the backend that ships
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.tensor([steps], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        steps = int(t.item())
    return steps


def step_bounds(n_local: int, steps: int):
    """Split this rank's n_local items into exactly ``steps`` contiguous batches whose sizes differ by at most one
    (empty batches only when n_local < steps: the rank then runs a zero-gradient step and still joins the exchange)."""
    return [shard_range(n_local, i, steps) for i in range(steps)] if steps > 0 else []


class GradAllReduce:
    """Sum a flat gradient buffer across ranks.

    Payload at the DB15K shape is ~10 MB (7.7 MB entity table + projections + Mformer): on 8 fully
    connected MI355X (7 xGMI links x ~153 GB/s each) that is latency-bound, so it goes out as ONE
    collective on the compute stream's successor rather than per-tensor buckets.  `bucket_bytes`
    splits larger payloads (config 5: 512 MB table) so that reduce-scatter/all-gather phases of
    consecutive buckets overlap on the links.
    """

    def __init__(self, world: int, bucket_bytes: int = 256 << 20, group=None, force: bool = False):
        self.world = world
        self.force = force          # run the collectives even with one rank (exercises the path on a 1-GPU box)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.group = group

    # --- overlapped form: start() a bucket as soon as its gradients are final, finish() before the optimiser ------------
    def start(self, bucket: torch.Tensor):
        """Enqueue the summed all-reduce of one contiguous slice of the flat gradient buffer without blocking the
        compute stream: the collective runs on the backend's own stream behind everything enqueued so far, while the
        kernels launched after this call (the rest of the backward) keep the CUs busy."""
        if self.world <= 1 and not self.force:
            return
        if not hasattr(self, "_pending"):
            self._pending = []
        n = bucket.numel()
        for lo in range(0, n, self.bucket_elems):
            self._pending.append(dist.all_reduce(bucket[lo:lo + self.bucket_elems], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def start_gather(self, full: torch.Tensor, shard: RowShard):
        """A gradient region in which every rank wrote ONLY the rows it owns (the entity table under row-sharded fusion): an
        in-place all-gather moves 1/world of what the summed all-reduce of the same region would.  ``full`` is the region padded
        to whole shards, [shard.n_pad, D] (the engine reserves the pad rows inside its flat buffer).  A region WITHOUT the pad rows
        (n_rows not divisible by world) cannot be gathered in place: its foreign rows are cleared first -- they hold what the
        previous exchange left there -- and it goes through the summed all-reduce."""
        if self.world <= 1 and not self.force:
            return
        if full.shape[0] != shard.n_pad:
            if full.shape[0] != shard.n_rows:
                raise ValueError(f"start_gather: region of {full.shape[0]} rows, shards cover {shard.n_rows} (padded {shard.n_pad})")
            full[:shard.lo].zero_()
            full[shard.lo + shard.n_loc:].zero_()
            return self.start(full.reshape(-1))
        if not hasattr(self, "_pending"):
            self._pending = []
        mine = full[shard.lo:shard.lo + shard.n_shard]
        if full.is_cuda and dist.get_backend(self.group) == "gloo":      # test transport only, as in RowShard.all_gather
            host = torch.empty(full.shape, dtype=full.dtype)
            dist.all_gather_into_tensor(host, mine.cpu(), group=self.group)
            full.copy_(host)
            return
        self._pending.append(dist.all_gather_into_tensor(full, mine, group=self.group, async_op=True))

    def finish(self):
        """Make the compute stream wait for every started bucket (no host synchronisation on the GPU backends)."""
        for w in getattr(self, "_pending", []):
            w.wait()
        self._pending = []

    def __call__(self, flat: torch.Tensor):
        if self.world <= 1 and not self.force:
            return flat
        n = flat.numel()
        if n <= self.bucket_elems:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            return flat
        works = []
        for lo in range(0, n, self.bucket_elems):
            works.append(dist.all_reduce(flat[lo:lo + self.bucket_elems], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()
        return flat
