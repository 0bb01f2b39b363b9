"""N>1 path on CPU: world_size-2 gloo run of the cross-stream scanner/activity reduction
(the only collective of the path)."""

import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank: int, world: int, port: int, out_dir: str):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "wavecap-sdr_amd")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wavehip.scanner_reduce import (activity_from_stats, best_channel, gather_measurements,
                                        reduce_channel_stats)

    rng = np.random.default_rng(100 + rank)
    M, hops = 64, 50 + 10 * rank
    p = rng.random((hops, M)) ** 2
    p[:, 7 + rank] += 5.0                                   # one active channel per stream
    stats = torch.from_numpy(np.stack([p.sum(0), (p * p).sum(0), np.full(M, float(hops)), p.min(0), p.max(0)], 1))
    merged = reduce_channel_stats(stats.clone())
    meas = torch.tensor([[10.0 * rank, 3.0 + rank, float(rank == 1)]], dtype=torch.float64)
    allm = gather_measurements(meas)
    active, rssi = activity_from_stats(merged, squelch_db=0.0)
    torch.save(dict(stats=stats, merged=merged, allm=allm, active=active,
                    best=best_channel(allm[:, 0, 1], allm[:, 0, 2])), os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_scanner_reduce_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), f"r{i}.pt"), weights_only=True) for i in range(world)]
    exp = torch.cat([r[0]["stats"][:, :3] + r[1]["stats"][:, :3],
                     torch.minimum(r[0]["stats"][:, 3], r[1]["stats"][:, 3])[:, None],
                     torch.maximum(r[0]["stats"][:, 4], r[1]["stats"][:, 4])[:, None]], 1)
    for i in range(world):
        assert torch.allclose(r[i]["merged"], exp, rtol=1e-12, atol=0)
        assert r[i]["allm"].shape == (2, 1, 3) and r[i]["allm"][1, 0, 0] == 10.0
        assert r[i]["best"] == 1                            # rank 1 is the synced candidate
        act = torch.nonzero(r[i]["active"]).flatten().tolist()
        assert act == [7, 8]


def test_reduce_is_identity_without_process_group():
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "wavecap-sdr_amd")]
    from wavehip.scanner_reduce import reduce_channel_stats

    s = torch.rand(8, 5, dtype=torch.float64)
    assert reduce_channel_stats(s) is s


def _worker_async(rank: int, world: int, port: int, out_dir: str):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "wavecap-sdr_amd")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wavehip.scanner_reduce import AsyncStatsReducer

    red = AsyncStatsReducer()
    outs = []
    for i in range(3):
        s = torch.full((16, 5), float(rank + 1 + 10 * i), dtype=torch.float64)
        prev = red.wait()
        red.submit(s)
        if prev is not None:
            outs.append(prev)
    outs.append(red.wait())
    torch.save(outs, os.path.join(out_dir, f"a{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_async_stats_reducer_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker_async, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        outs = torch.load(os.path.join(str(tmp_path), f"a{r}.pt"), weights_only=True)
        assert len(outs) == 3
        for i, m in enumerate(outs):
            a, b = 1.0 + 10 * i, 2.0 + 10 * i
            assert torch.equal(m[:, 0:3], torch.full((16, 3), a + b, dtype=torch.float64))
            assert torch.equal(m[:, 3], torch.full((16,), a, dtype=torch.float64))
            assert torch.equal(m[:, 4], torch.full((16,), b, dtype=torch.float64))
