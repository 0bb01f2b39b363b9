"""Multi-GPU readiness without multi-GPU hardware (two gloo ranks on the CPU): the bench's N > 1 control flow
(bench.drive: matched step counts, the reducer's buffer reuse), the single-stream channel split (partition / broadcast /
gather / merge), and the reducer's independence from the caller's buffer."""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd"), os.path.join(ROOT, "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _worker_drive(rank, world, port, out_dir):
    _init(rank, world, port)
    import bench
    from wavehip.scanner_reduce import AsyncStatsReducer

    M = 32
    red = AsyncStatsReducer()
    stats = torch.zeros((M, 5), dtype=torch.float64)     # ONE buffer, rewritten every step (the reducer must not alias it)
    it = [0]
    seen = []

    def step():
        it[0] += 1
        v = float(100 * rank + it[0])
        stats[:, 0:3] = v
        stats[:, 3] = v
        stats[:, 4] = v
        g = red.wait(merge=False)
        red.submit(stats)
        stats.fill_(-1.0)                                  # overwritten right after submit: must not reach the other rank
        if g is not None:
            seen.append(g.clone())

    elapsed, last = bench.drive(step, red, steps=4, warmup=2, prewarm=5, sync=lambda: None, barrier=dist.barrier)
    torch.save(dict(last=last, n_seen=len(seen), steps=it[0], submitted=red.submitted, seen_ok=all(bool((g >= 0).all()) for g in seen)),
               os.path.join(out_dir, f"d{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_control_flow_world2(tmp_path):
    """bench.drive on two ranks: 5 + 2 + 4 steps each, one all-gather per step, the last scan collected inside the timed
    region; every rank ends with the merge of both ranks' LAST step and never sees the -1 the caller wrote into its own
    buffer after submit()."""
    world, port = 2, _free_port()
    mp.spawn(_worker_drive, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        d = torch.load(os.path.join(str(tmp_path), f"d{r}.pt"), weights_only=True)
        assert d["steps"] == 11 and d["submitted"] == 11 and d["n_seen"] == 10 and d["seen_ok"]
        a, b = 11.0, 111.0                                # step 11 of rank 0 and of rank 1
        assert torch.equal(d["last"][:, 0:3], torch.full((32, 3), a + b, dtype=torch.float64))
        assert torch.equal(d["last"][:, 3], torch.full((32,), a, dtype=torch.float64))
        assert torch.equal(d["last"][:, 4], torch.full((32,), b, dtype=torch.float64))


def _worker_split(rank, world, port, out_dir):
    _init(rank, world, port)
    from wavehip import channel_split as cs

    cfgs = [dict(id=f"ch{k}", offset=1000.0 * k) for k in range(7)]
    calls = []

    def compute(samples, mine):
        calls.append([c["id"] for c in mine])
        return [(c["id"], float(samples.sum()) + c["offset"]) for c in mine]

    x = torch.arange(1000, dtype=torch.float32) if rank == 0 else None
    res = cs.split_process(x, cfgs, compute, src=0)
    res2 = cs.split_process(torch.ones(10) if rank == 0 else torch.empty(10), cfgs[:1], compute, src=0)   # fewer channels than ranks
    torch.save(dict(res=res, res2=res2, calls=calls), os.path.join(out_dir, f"s{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_channel_split_world2(tmp_path):
    from importlib import import_module
    sys.path[:0] = [os.path.join(ROOT, "wavecap-sdr_amd")]
    cs = import_module("wavehip.channel_split")
    assert cs.partition(7, 2) == [(0, 4), (4, 7)] and cs.partition(1, 2) == [(0, 1), (1, 1)]
    assert cs.partition(32, 8) == [(4 * r, 4 * r + 4) for r in range(8)] and cs.partition(0, 3) == [(0, 0)] * 3
    with pytest.raises(RuntimeError):
        cs.merge([[1, 2], [3]], 4)
    world, port = 2, _free_port()
    mp.spawn(_worker_split, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), "s0.pt"), weights_only=False)
    r1 = torch.load(os.path.join(str(tmp_path), "s1.pt"), weights_only=False)
    total = float(np.arange(1000, dtype=np.float32).sum())
    assert r0["res"] == [(f"ch{k}", total + 1000.0 * k) for k in range(7)]       # channel order, every channel once
    assert r1["res"] is None
    assert r0["calls"][0] == ["ch0", "ch1", "ch2", "ch3"] and r1["calls"][0] == ["ch4", "ch5", "ch6"]
    assert r0["res2"] == [("ch0", 10.0)] and len(r1["calls"]) == 1                # rank 1 had nothing to do the second time
