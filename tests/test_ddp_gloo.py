"""CPU, world_size 2, gloo: the data-parallel gradient exchange of
tiny_diffusion_amd.train (bucketed async all-reduce of slices of the flat gradient)
gives the mean of the per-rank gradients, every element exactly once."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tiny_diffusion_amd.conditional_diffusion import NoiseModel
        from tiny_diffusion_amd.train import BucketedAllReduce, plan_buckets

        m = NoiseModel()
        named = dict(m.named_parameters())
        offsets, o = {}, 0
        for n in m._param_order:
            offsets[n] = (o, o + named[n].numel())
            o += named[n].numel()
        buckets = plan_buckets(offsets, True, 1 << 20)
        # every gradient element belongs to exactly one bucket
        cover = torch.zeros(o, dtype=torch.int32)
        for _, ranges in buckets:
            for lo, hi in ranges:
                cover[lo:hi] += 1
        assert int(cover.min()) == 1 and int(cover.max()) == 1
        assert [b[0] for b in buckets] == sorted(b[0] for b in buckets) and buckets[-1][0] == 14
        g = torch.Generator().manual_seed(100 + rank)
        flat = torch.randn(o, generator=g)
        mine = flat.clone()
        red = BucketedAllReduce(flat, buckets)
        assert red.world == world
        for bi in range(len(buckets)):
            red.launch(bi)
        scale = red.finish()
        assert scale == 1.0 / world
        others = [torch.randn(o, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
        want = sum(others) / world
        assert torch.allclose(flat * scale, want, rtol=1e-6, atol=1e-6)
        assert torch.equal(others[rank], mine)
        out[rank] = "ok"
    finally:
        torch.distributed.destroy_process_group()


def test_bucketed_allreduce_gloo_world2():
    world = 2
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: "ok", 1: "ok"}
