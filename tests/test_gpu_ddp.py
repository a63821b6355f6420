"""GPU, two ranks on one MI355X (gloo carries the collective; on a multi-GPU node the same code runs
over RCCL): SURVEY.md 8(e) parity definition for the data-parallel step - the all-reduced gradient
equals the mean over ranks of the reference gradient computed independently on each rank's shard
(train-mode BatchNorm statistics are rank-local), and all replicas hold identical parameters after
the optimizer step."""
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


def _is_pre_bn_bias(key):
    stage, idx, kind = (key.split(".") + ["", ""])[:3]
    return kind == "bias" and idx in ("0", "3") and stage[:3] in ("enc", "dec", "bot")


def _shard(rank, B=8):
    g = torch.Generator().manual_seed(500 + rank)
    x0 = torch.rand(B, 1, 28, 28, generator=g) * 2 - 1
    noise = torch.randn(B, 1, 28, 28, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    y = torch.randint(0, 10, (B,), generator=g)
    return x0, noise, t, y


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ref_cpu as R
        from oracle.weights import make_state_dict
        from tiny_diffusion_amd.conditional_diffusion import ForwardProcess, NoiseModel
        from tiny_diffusion_amd.train import TrainStep

        torch.cuda.set_device(0)
        sd = make_state_dict(6, True)
        m = NoiseModel()
        m.load_state_dict(sd)
        m = m.cuda().train()
        fp = ForwardProcess()
        ts = TrainStep(m, fp, lr=1e-3)
        assert ts.world == world
        ts.broadcast_parameters(0)
        x0, noise, t, y = _shard(rank)
        ts.step(x0.cuda(), y.cuda(), t=t.cuda(), noise=noise.cuda())
        torch.cuda.synchronize()
        got = ts.flat_grad.cpu() / world  # Adam folds the 1/world; the buffer holds the sum
        # reference: mean over ranks of the oracle gradient of each shard
        want = None
        for r in range(world):
            xr, nr, tr, yr = _shard(r)
            x_t = R.q_sample(R.Schedule(), xr, tr, nr)
            _, _, grads, _ = R.train_step_grads(sd, x_t, tr, nr, yr)
            want = grads if want is None else {k: want[k] + grads[k] for k in grads}
        want = {k: v / world for k, v in want.items()}
        bad = []
        for k, (lo, hi) in ts.offsets.items():
            if _is_pre_bn_bias(k):
                continue
            a, b = got[lo:hi].double(), want[k].reshape(-1).double()
            err = (a - b).norm().item() / max(b.norm().item(), 1e-30)
            # loose on purpose: this test is about the exchange (sum over ranks, 1/world, every
            # element once); fp32 noise of small-batch train-mode BN and near-tie max-pool routing
            # (up to ~1e-2 in the deep layers at B=8) is calibrated in test_gpu_unet.py
            if err > 3e-2:
                bad.append((k, err))
        assert not bad, bad
        # replicas stay identical: compare a checksum of the updated parameters across ranks
        probe = torch.stack([ts.flat_param.double().sum(), ts.flat_param.double().pow(2).sum()]).cpu()
        lo_, hi_ = probe.clone(), probe.clone()
        torch.distributed.all_reduce(lo_, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi_, op=torch.distributed.ReduceOp.MAX)
        assert torch.equal(lo_, hi_)
        out[rank] = "ok"
    finally:
        torch.distributed.destroy_process_group()


def test_data_parallel_step_matches_mean_of_shard_gradients():
    world = 2
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: "ok", 1: "ok"}
