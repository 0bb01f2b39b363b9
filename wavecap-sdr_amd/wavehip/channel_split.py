"""One device stream, several GPUs: split its CHANNEL SET (last paragraph).

When a node serves a single capture (one IQ stream) the per-stream sharding of config 5 leaves N-1 GPUs idle; the
channels of that stream are independent given the shared chunk, so they are what shards: rank r takes a contiguous range
of the capture's channel list, the chunk is broadcast (8 B per sample over xGMI: 80 MB/s at 10 MS/s, 19 MB/s at
2.4 MS/s -- negligible next to the link's ~150 GB/s), every rank runs its own ChannelBank / filterbank on the whole
chunk, and the small per-channel results (audio: 2 400 float32 per channel and chunk; activity statistics: 40 B per
channel) return to the source rank with one gather.  The scanner / activity reduction is unchanged (it merges
per-channel statistics, whoever computed them).

    parts = partition(n_channels, world)            -> [(lo, hi)] per rank, sizes differ by at most one
    split_process(samples, cfgs, compute, src=0)    -> on src: [result per channel in cfg order]; elsewhere: None

`compute(samples, cfgs_of_this_rank) -> list of per-channel results` is the rank-local work (ChannelDispatcher.process on
a GPU box).  Collectives: one broadcast of the chunk + one gather_object of the result lists per chunk; with backend
"nccl" the broadcast runs on RCCL over xGMI (device tensor), the result lists are host objects.
"""

from __future__ import annotations


def partition(n_items: int, world: int) -> list[tuple[int, int]]:
    """Contiguous ranges, sizes differ by at most one, the first n_items % world ranks get the extra item (ranks
    beyond the item count get an empty range)."""
    base, extra = divmod(max(0, int(n_items)), max(1, int(world)))
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def merge(parts_results: list[list], n_items: int) -> list:
    """Results of all ranks (rank order) -> one list in channel order; checks that nothing is missing or duplicated."""
    flat = [r for part in parts_results for r in part]
    if len(flat) != n_items:
        raise RuntimeError(f"channel split: {len(flat)} results for {n_items} channels")
    return flat


def split_process(samples, cfgs, compute, src: int = 0, group=None):
    """See the module docstring.  `samples` is needed on `src` only (a torch tensor, host or device; other ranks pass a
    tensor of the same shape / dtype / device to receive into, or None to have one allocated from the broadcast header)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return compute(samples, list(cfgs))
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    cfgs = list(cfgs)
    # 1. the chunk: header (shape, dtype) as an object, then the payload as one tensor broadcast
    hdr = [None]
    if rank == src:
        hdr[0] = (tuple(samples.shape), str(samples.dtype).replace("torch.", ""), str(samples.device.type))
    dist.broadcast_object_list(hdr, src=src, group=group)
    shape, dtype, _ = hdr[0]
    if rank != src and samples is None:
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        samples = torch.empty(shape, dtype=getattr(torch, dtype), device=dev)
    dist.broadcast(samples, src=src, group=group)
    # 2. the rank's share of the channel list
    lo, hi = partition(len(cfgs), world)[rank]
    mine = compute(samples, cfgs[lo:hi]) if hi > lo else []
    if len(mine) != hi - lo:
        raise RuntimeError(f"channel split: rank {rank} produced {len(mine)} results for {hi - lo} channels")
    # 3. results back to the source
    gathered = [None] * world if rank == src else None
    dist.gather_object(mine, gathered, dst=src, group=group)
    return merge(gathered, len(cfgs)) if rank == src else None
