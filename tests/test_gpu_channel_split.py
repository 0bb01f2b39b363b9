"""One capture's channel set split over two ranks with the REAL ChannelDispatcher (wavehip.channel_split.
split_process_device; the shape it replaces: capture.py:2489-2597, one Capture feeding all its channels).  Two fresh child
processes (gloo, sharing cuda:0 -- a one-GPU box has no second card and RCCL needs one GPU per rank) run the split on a
chunk of 2.4 MS/s IQ with 7 channels of three chain kinds; the source rank's gathered device rows must equal the
single-rank dispatcher's results bit for bit, in channel order.  The collectives move tensors only (no pickled result
lists); with backend "nccl" the same calls run on RCCL between device buffers."""

import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfgs(wavehip, offs):
    kinds = [dict(mode="nbfm", enable_deemphasis=False), dict(mode="nbfm"), dict(mode="wbfm"),
             dict(mode="nbfm", enable_deemphasis=False), dict(mode="nbfm", enable_deemphasis=False, enable_noise_reduction=True),
             dict(mode="wbfm"), dict(mode="nbfm")]
    return [wavehip.ChannelConfig(id=f"c{k}", offset_hz=offs[k], **kw) for k, kw in enumerate(kinds)]


def _worker(rank, world, port, out_dir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd"), os.path.join(ROOT, "tests")]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    import signals as S
    import wavehip
    from wavehip import channel_split as cs

    fs, n = 2_400_000, 120_000
    offs = S.nbfm_bank_offsets(8)
    cfgs = _cfgs(wavehip, offs)
    disp = wavehip.ChannelDispatcher(fs)
    for chunk in range(2):          # second chunk: banks reused, offsets retuned on the fly
        if chunk == 1:
            for c in cfgs:
                c.offset_hz += 2000.0
        d = torch.from_numpy(S.nbfm_bank_c64(n, fs, seed=51 + chunk, n_ch=8)).cuda() if rank == 0 else None
        rows = cs.split_process_device(d, cfgs, disp, src=0, n_samples=n)
        if rank == 0:
            assert rows.is_cuda and rows.shape[0] == len(cfgs)
            torch.save(rows.cpu(), os.path.join(out_dir, f"rows{chunk}.pt"))
        else:
            assert rows is None
    torch.save(dict(banks=disp.banks_created), os.path.join(out_dir, f"info{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_channel_split_with_the_real_dispatcher_two_ranks(tmp_path):
    import torch
    import torch.multiprocessing as mp
    import signals as S
    import wavehip

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    fs, n = 2_400_000, 120_000
    offs = S.nbfm_bank_offsets(8)
    cfgs = _cfgs(wavehip, offs)
    single = wavehip.ChannelDispatcher(fs)
    for chunk in range(2):
        if chunk == 1:
            for c in cfgs:
                c.offset_hz += 2000.0
        iq = S.nbfm_bank_c64(n, fs, seed=51 + chunk, n_ch=8)
        want = single.process(iq, cfgs)
        rows = torch.load(os.path.join(str(tmp_path), f"rows{chunk}.pt"), weights_only=True)
        got = wavehip.ChannelDispatcher.rows_to_results(rows)
        assert len(got) == len(want) == 7
        for k, ((a, m), (wa, wm)) in enumerate(zip(got, want)):
            assert (a is None) == (wa is None) and m == wm, (chunk, k)
            if a is not None:
                assert a.shape == wa.shape and np.array_equal(a, wa), (chunk, k)
        # the device-resident single-rank form gives the same rows
        own = wavehip.ChannelDispatcher.rows_to_results(single.process_device(torch.from_numpy(iq).cuda(), cfgs))
        assert all(np.array_equal(x[0], y[0]) and x[1] == y[1] for x, y in zip(own, want))
    # rank 0 had channels 0..3 (nbfm plain x2, nbfm de-emphasis, wbfm), rank 1 channels 4..6: banks built once per chain
    i0 = torch.load(os.path.join(str(tmp_path), "info0.pt"), weights_only=True)
    i1 = torch.load(os.path.join(str(tmp_path), "info1.pt"), weights_only=True)
    assert i0["banks"] == 3 and i1["banks"] == 3
