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

    split_process_device(d_samples, cfgs, dispatcher, src=0)
                                                    -> on src: float32 tensor [n_channels, row_len + 5] (device), else None

`compute(samples, cfgs_of_this_rank) -> list of per-channel results` is the rank-local work of the generic form
(collectives: one broadcast of the chunk + one gather_object of the result lists per chunk -- host objects).
`split_process_device` is the device-resident form around a real ChannelDispatcher: the chunk is broadcast as a device
tensor, every rank's share comes back from `dispatcher.process_device` as one float32 row block, one all-reduce(MAX) of
a single integer agrees on the row length and ONE `dist.gather` of equal-sized tensors returns the blocks to the source
rank -- no pickling, nothing through host objects; with backend "nccl" all three run on RCCL over xGMI between device
buffers (with "gloo" -- the CPU rehearsal -- the tensors are staged through the host inside the helper).
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


def _coll_tensor(t, group):
    """gloo (the rehearsal backend) moves host tensors; nccl / RCCL moves device tensors in place."""
    import torch.distributed as dist

    return t.cpu() if dist.get_backend(group) == "gloo" and t.is_cuda else t


def split_process_device(d_samples, cfgs, dispatcher, src: int = 0, group=None, n_samples: int | None = None):
    """One chunk of one capture, its channels split over the ranks, results kept on the device.

    d_samples: complex64 device tensor [n] on `src` (other ranks: None, or a tensor to receive into; then `n_samples`
    or the tensor gives the length).  cfgs: the capture's channel configs, the same list on every rank (the control
    plane hands every rank the capture's channel table).  dispatcher: this rank's ChannelDispatcher.
    Returns on src a float32 device tensor [len(cfgs), row_len + 5] in channel order (ChannelDispatcher.process_device
    layout; `ChannelDispatcher.rows_to_results` turns it into the reference's tuples), None elsewhere."""
    import torch
    import torch.distributed as dist

    cfgs = list(cfgs)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return dispatcher.process_device(d_samples, cfgs)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.device("cuda", torch.cuda.current_device())
    # 1. the chunk (length known to every rank: the capture's chunk size, capture.py:3035)
    if rank != src and d_samples is None:
        if n_samples is None:
            raise ValueError("split_process_device: ranks other than src need n_samples or a receive tensor")
        d_samples = torch.empty(int(n_samples), dtype=torch.complex64, device=dev)
    buf = _coll_tensor(torch.view_as_real(d_samples), group)
    dist.broadcast(buf, src=src, group=group)
    if buf.device != d_samples.device:
        d_samples = torch.view_as_complex(buf.to(dev))
    # 2. this rank's share
    lo, hi = partition(len(cfgs), world)[rank]
    kmax = partition(len(cfgs), world)[0][1]
    mine = dispatcher.process_device(d_samples, cfgs[lo:hi]) if hi > lo else None
    # 3. one row length for all ranks, then one gather of equal-sized blocks
    L = torch.tensor([0 if mine is None else mine.shape[1] - dispatcher.ROW_EXTRA], dtype=torch.int64, device=dev)
    Lc = _coll_tensor(L, group)
    dist.all_reduce(Lc, op=dist.ReduceOp.MAX, group=group)
    L = int(Lc.item())
    E = dispatcher.ROW_EXTRA
    block = torch.zeros((kmax, L + E), dtype=torch.float32, device=dev)
    if mine is not None:
        own = mine.shape[1] - E
        block[:hi - lo, :own] = mine[:, :own]
        block[:hi - lo, L:] = mine[:, own:]
    send = _coll_tensor(block, group)
    recv = [torch.empty_like(send) for _ in range(world)] if rank == src else None
    dist.gather(send, recv, dst=src, group=group)
    if rank != src:
        return None
    parts = partition(len(cfgs), world)
    rows = torch.cat([recv[r][:parts[r][1] - parts[r][0]] for r in range(world)], dim=0)
    if rows.shape[0] != len(cfgs):
        raise RuntimeError(f"channel split: {rows.shape[0]} rows for {len(cfgs)} channels")
    return rows.to(dev)
