"""GPU, two ranks: SURVEY.md 8(e) parity definition for the data-parallel step - the all-reduced
gradient equals the mean over ranks of the reference gradient computed independently on each
rank's shard (train-mode BatchNorm statistics are rank-local), and all replicas hold identical
parameters after the optimizer step.

Two variants of the same worker: gloo carrying the collective with both ranks on ONE MI355X (runs
on the 1-GPU box), and RCCL ("nccl") with one rank per GPU (runs wherever >= 2 GPUs are visible,
skips otherwise).

Gate: the single-GPU gradient gate of test_gpu_unet.py.  Every rank evaluates the oracle on ITS shard
in fp32 and in fp64 with the sub-gradient choices its GPU forward made - the max-pool routing and
the ReLU active sets (a window tie or an activation within fp32 rounding of 0 is a coin flip, and
at B = 8 per shard ONE flipped ReLU in a deep layer moves that layer's gradient by 5e-3 through
train-mode BatchNorm: measured in round 2, tools/gpu_ddp_debug.py - with the GPU's choices the fp64
oracle agrees with the GPU to 5e-6).  The per-shard oracle gradients are averaged over ranks, and
the all-reduced GPU gradient must be as close to the fp64 mean as 10x the fp32 oracle's own distance
from it (floor 1e-4 per parameter)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard(rank, B=8):
    g = torch.Generator().manual_seed(500 + rank)
    x0 = torch.rand(B, 1, 28, 28, generator=g) * 2 - 1
    noise = torch.randn(B, 1, 28, 28, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    y = torch.randint(0, 10, (B,), generator=g)
    return x0, noise, t, y


def _worker(rank, world, port, out, backend):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev_index = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev_index)
    if backend == "nccl":
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world,
                                             device_id=torch.device("cuda", dev_index))
        cpu_group = torch.distributed.new_group(backend="gloo")   # carries the oracle's CPU tensors
    else:
        torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
        cpu_group = None
    try:
        from oracle import ref_cpu as R
        from oracle.weights import make_state_dict
        from parity_helpers import gpu_pool_routing, gpu_relu_masks, grad_precision_failures
        from tiny_diffusion_amd.conditional_diffusion import ForwardProcess, NoiseModel
        from tiny_diffusion_amd.train import TrainStep

        B = 8
        sd = make_state_dict(6, True)
        m = NoiseModel()
        m.load_state_dict(sd)
        m = m.cuda().train()
        fp = ForwardProcess()
        ts = TrainStep(m, fp, lr=1e-3)
        assert ts.world == world
        ts.broadcast_parameters(0)
        x0, noise, t, y = _shard(rank, B)
        ts.step(x0.cuda(), y.cuda(), t=t.cuda(), noise=noise.cuda())
        torch.cuda.synchronize()
        got = ts.flat_grad.cpu() / world  # Adam folds the 1/world; the buffer holds the sum
        # this rank's shard on the oracle, routed like this rank's GPU forward
        x_t = R.q_sample(R.Schedule(), x0, t, noise)
        cpu_args = (sd, x_t, t, noise, y)
        pidx = gpu_pool_routing(m, B, cpu_args)
        masks, _ = gpu_relu_masks(m, B, cpu_args, pool_idx=pidx)
        _, _, g32, _ = R.train_step_grads(*cpu_args, pool_idx=pidx, relu_masks=masks)
        _, _, g64, _ = R.train_step_grads(*cpu_args, dtype=torch.float64, pool_idx=pidx, relu_masks=masks)
        names = list(ts.offsets)
        f32 = torch.cat([g32[k].reshape(-1).double() for k in names])
        f64 = torch.cat([g64[k].reshape(-1) for k in names])
        torch.distributed.all_reduce(f32, group=cpu_group)
        torch.distributed.all_reduce(f64, group=cpu_group)
        f32 /= world
        f64 /= world
        m32, m64, mine, o = {}, {}, {}, 0
        for k in names:
            n = g64[k].numel()
            m32[k], m64[k] = f32[o:o + n].view(g64[k].shape), f64[o:o + n].view(g64[k].shape)
            lo, hi = ts.offsets[k]
            mine[k] = got[lo:hi].view(g64[k].shape)
            o += n
        bad = grad_precision_failures(mine, m32, m64, True)
        assert not bad, bad
        # replicas stay identical: compare a checksum of the updated parameters across ranks
        probe = torch.stack([ts.flat_param.double().sum(), ts.flat_param.double().pow(2).sum()]).cpu()
        lo_, hi_ = probe.clone(), probe.clone()
        torch.distributed.all_reduce(lo_, op=torch.distributed.ReduceOp.MIN, group=cpu_group)
        torch.distributed.all_reduce(hi_, op=torch.distributed.ReduceOp.MAX, group=cpu_group)
        assert torch.equal(lo_, hi_)
        # in-kernel noise: ranks built with the SAME seed must still draw different noise
        ts2 = TrainStep(NoiseModel().cuda().train(), fp, philox_seed=1234)
        _, n_mine = fp.q_sample_philox(x0.cuda(), t.cuda(), ts2.philox_seed, ts2._philox_offset())
        n_all = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        torch.distributed.all_gather(n_all, n_mine.double().sum().cpu().view(1), group=cpu_group)
        assert len({float(v) for v in n_all}) == world, n_all
        out[rank] = "ok"
    finally:
        torch.distributed.destroy_process_group()


def _syncbn_worker(rank, world, port, out, backend):
    """SyncBN: with BatchNorm statistics (forward) and the backward sums taken over the GLOBAL batch, the
    two-rank step must equal the single-process step on the concatenated batch - the exact-parity mode
    SURVEY.md 8(e) names: all-reduced gradient / world == oracle gradient at B = 2 x 8, identical running
    statistics on both ranks == the oracle's for the global batch."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev_index = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev_index)
    if backend == "nccl":
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world,
                                             device_id=torch.device("cuda", dev_index))
        cpu_group = torch.distributed.new_group(backend="gloo")
    else:
        torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
        cpu_group = None
    try:
        from oracle import ref_cpu as R
        from oracle.weights import make_state_dict
        from parity_helpers import check_choices, gpu_pool_idx_raw, gpu_relu_masks_raw, grad_precision_failures
        from tiny_diffusion_amd.conditional_diffusion import ForwardProcess, NoiseModel
        from tiny_diffusion_amd.train import TrainStep

        B = 8
        sd = make_state_dict(6, True)
        m = NoiseModel()
        m.load_state_dict(sd)
        m = m.cuda().train()
        fp = ForwardProcess()
        ts = TrainStep(m, fp, lr=1e-3, sync_bn=True)
        assert ts.world == world and ts.sync_bn
        ts.broadcast_parameters(0)
        x0, noise, t, y = _shard(rank, B)
        ts.step(x0.cuda(), y.cuda(), t=t.cuda(), noise=noise.cuda())
        torch.cuda.synchronize()
        assert m._bn_error is None, m._bn_error
        got = ts.flat_grad.cpu() / world
        # the global batch = the shards in rank order; the sub-gradient choices of each rank's GPU run
        shards = [_shard(r, B) for r in range(world)]
        X0, N, T, Y = (torch.cat([s_[i] for s_ in shards]) for i in range(4))
        X_t = R.q_sample(R.Schedule(), X0, T, N)
        pool, masks = gpu_pool_idx_raw(m, B), gpu_relu_masks_raw(m, B)

        def gather(v):   # (B, ...) per rank -> (world * B, ...) in rank order
            parts = [torch.zeros_like(v) for _ in range(world)]
            torch.distributed.all_gather(parts, v.contiguous(), group=cpu_group)
            return torch.cat(parts)

        pool = {k: gather(v) for k, v in pool.items()}
        masks = {k: gather(v.to(torch.uint8)).bool() for k, v in masks.items()}
        cpu_args = (sd, X_t, T, N, Y)
        flips = check_choices(pool, masks, cpu_args)          # only ties may differ from the exact choices
        kw = dict(pool_idx=pool, relu_masks=masks)
        _, _, g32, bufs = R.train_step_grads(*cpu_args, **kw)
        _, _, g64, _ = R.train_step_grads(*cpu_args, dtype=torch.float64, **kw)
        mine = {k: got[lo:hi].view(g64[k].shape) for k, (lo, hi) in ts.offsets.items()}
        bad = grad_precision_failures(mine, g32, g64, True)
        assert not bad, (flips, bad)
        for k, v in m.state_dict().items():   # running statistics of the GLOBAL batch, on every rank
            if "running_" in k:
                assert torch.allclose(v.cpu(), bufs[k], rtol=2e-5, atol=2e-5), k
            if "num_batches" in k:
                assert int(v) == int(bufs[k])
        # and the default (rank-local statistics) really is something else: the two modes differ
        m2 = NoiseModel(); m2.load_state_dict(sd); m2 = m2.cuda().train()
        ts2 = TrainStep(m2, fp, lr=1e-3)
        ts2.step(x0.cuda(), y.cuda(), t=t.cuda(), noise=noise.cuda())
        torch.cuda.synchronize()
        assert (ts2.flat_grad.cpu() / world - got).norm() > 1e-3 * got.norm()
        out[rank] = "ok"
    finally:
        torch.distributed.destroy_process_group()


def _run(backend, worker=None):
    world = 2
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(worker or _worker, args=(world, port, out, backend), nprocs=world, join=True)
    assert dict(out) == {0: "ok", 1: "ok"}


def test_data_parallel_step_matches_mean_of_shard_gradients():
    _run("gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL variant needs two GPUs")
def test_data_parallel_step_rccl_two_gpus():
    _run("nccl")


def test_sync_batchnorm_two_ranks_equal_the_global_batch_step():
    _run("gloo", _syncbn_worker)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL variant needs two GPUs")
def test_sync_batchnorm_rccl_two_gpus():
    _run("nccl", _syncbn_worker)
