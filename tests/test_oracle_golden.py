"""CPU: the oracle (oracle/ref_cpu.py) against the golden vectors generated
from the reference itself (tools/make_golden.py).  This is what pins parity:
the GPU tests compare the HIP path with this oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from oracle.weights import make_state_dict, state_dict_sha256, key_shapes

torch.set_num_threads(min(8, os.cpu_count() or 1))

# fp32 noise floor of the reference against itself in fp64 is 3-5e-13 relative
# MSE; the oracle must sit at that floor.
REL_MSE_TOL = 1e-10


def rel_mse(a, b):
    a, b = a.double(), b.double()
    return ((a - b) ** 2).mean().item() / max((b**2).mean().item(), 1e-30)


def is_pre_bn_bias(key):
    stage, idx, kind = (key.split(".") + ["", ""])[:3]
    return kind == "bias" and idx in ("0", "3") and stage[:3] in ("enc", "dec", "bot")


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def test_weight_generator_hash(golden_dir):
    want = dict(
        line.split()[0:3:2] for line in open(os.path.join(golden_dir, "weights.sha256"))
    )
    assert state_dict_sha256(make_state_dict(0, False)) == want["uncond"]
    assert state_dict_sha256(make_state_dict(0, True)) == want["cond"]


def test_state_dict_layout():
    # 105 entries unconditional (+1 class_embedding); the reference model accepts
    # exactly this key list with load_state_dict(strict=True) in tools/make_golden.py
    unc, con = key_shapes(False), key_shapes(True)
    assert len(unc) == 105 and len(con) == 106
    n_unc = sum(int(np.prod(s)) for k, s, kind in unc if not kind.startswith("bn_r") and kind != "bn_nbt")
    n_con = sum(int(np.prod(s)) for k, s, kind in con if not kind.startswith("bn_r") and kind != "bn_nbt")
    assert n_unc == 11_182_273 and n_con == 11_184_833


def test_schedule_bit_exact(golden_dir):
    d = load(golden_dir, "schedule")
    s = R.Schedule()
    c1, c2, sig = s.p_sample_coeffs()
    for name, v in [("betas", s.betas), ("alphas", s.alphas), ("alphas_cumprod", s.alphas_cumprod),
                    ("c1", c1), ("c2", c2), ("sigma", sig)]:
        assert np.array_equal(v.numpy(), d[name]), name
    # SURVEY.md 8(a) a1 spot values
    assert abs(s.alphas_cumprod[0].item() - 0.99989998) < 1e-7
    assert abs(s.alphas_cumprod[999].item() - 4.0358e-05) < 1e-8


@pytest.mark.parametrize(
    "name,cond,training",
    [
        ("fwd_B64_train_uncond", False, True),
        ("fwd_B64_eval_uncond", False, False),
        ("fwd_B16_train_cond", True, True),
        ("fwd_B16_eval_cond", True, False),
        ("fwd_B8_eval_rawt", False, False),
    ],
)
def test_forward_matches_reference(golden_dir, name, cond, training):
    d = load(golden_dir, name)
    sd = make_state_dict(int(d["seed"]), cond, time_scale=float(d["time_scale"]))
    p, b = R.split_state(sd)
    sched = R.Schedule()
    t = torch.from_numpy(d["t"])
    y = torch.from_numpy(d["y"]) if cond else None
    noise = torch.from_numpy(d["noise"])
    x_t = R.q_sample(sched, torch.from_numpy(d["x0"]), t, noise)
    assert torch.equal(x_t, torch.from_numpy(d["x_t"]))  # q_sample is bit-exact
    with torch.no_grad():
        eps = R.unet_forward(p, b, x_t, t, y, training=training)
    assert rel_mse(eps, torch.from_numpy(d["eps_hat"])) < REL_MSE_TOL
    loss = torch.nn.functional.mse_loss(eps, noise).item()
    assert abs(loss - float(d["loss"])) <= 1e-5 * abs(float(d["loss"]))
    if training:
        for k, v in b.items():
            ref = torch.from_numpy(d[k.replace(".", "__")])
            if v.dtype == torch.int64:
                assert int(v) == int(ref) == 2
            else:
                assert torch.allclose(v, ref, rtol=1e-5, atol=1e-5), k


def test_taps_match_reference(golden_dir):
    d = load(golden_dir, "taps_B2")
    sd = make_state_dict(int(d["seed"]), False)
    p, b = R.split_state(sd)
    taps = {}
    with torch.no_grad():
        R.unet_forward(p, b, torch.from_numpy(d["x"]), torch.from_numpy(d["t"]), None, False, taps)
    names = sorted({k.split("__")[0] for k in d.files if "__" in k})
    assert len(names) == 23
    for n in names:
        v = taps[n]
        assert v.numel() == int(np.prod(d[f"{n}__shape"])), n
        flat = v.contiguous().view(-1)
        got = flat[torch.from_numpy(d[f"{n}__idx"])]
        ref = torch.from_numpy(d[f"{n}__val"])
        scale = max(ref.abs().max().item(), 1e-6)
        assert (got - ref).abs().max().item() <= 2e-5 * scale, n
        asum = float(d[f"{n}__asum"])
        assert abs(flat.double().abs().sum().item() - asum) <= 1e-5 * asum + 1e-6, n


@pytest.mark.parametrize("name,cond", [("grad_B64_uncond", False), ("grad_B16_cond", True)])
def test_backward_and_adam_match_reference(golden_dir, name, cond):
    d = load(golden_dir, name)
    sd = make_state_dict(int(d["seed"]), cond)
    y = torch.from_numpy(d["y"]) if cond else None
    loss, eps, grads, bufs = R.train_step_grads(
        sd, torch.from_numpy(d["x_t"]), torch.from_numpy(d["t"]), torch.from_numpy(d["noise"]), y
    )
    assert abs(loss.item() - float(d["loss"])) <= 1e-5 * float(d["loss"])
    assert rel_mse(eps, torch.from_numpy(d["eps_hat"])) < REL_MSE_TOL
    params = {k: v.clone() for k, v in sd.items() if k in grads}
    for k, g in grads.items():
        kk = k.replace(".", "__")
        gn = float(d[f"gnorm__{kk}"])
        head = torch.from_numpy(d[f"ghead__{kk}"])
        got = g.contiguous().view(-1)
        if is_pre_bn_bias(k):
            # a bias in front of a train-mode BatchNorm has an exactly-zero true
            # gradient (the batch mean removes it); both sides hold rounding noise
            wn = float(d[f"gnorm__{kk.replace('bias', 'weight')}"])
            assert gn <= 1e-5 * wn and got.double().norm().item() <= 1e-5 * wn, k
            continue
        assert abs(got.double().norm().item() - gn) <= 5e-4 * gn, k
        rms = gn / np.sqrt(got.numel())
        assert (got[: head.numel()] - head).abs().max().item() <= 5e-3 * max(head.abs().max().item(), rms), k
    for k, v in bufs.items():
        ref = torch.from_numpy(d["buf__" + k.replace(".", "__")])
        if v.dtype != torch.int64:
            assert torch.allclose(v, ref, rtol=1e-5, atol=1e-5), k
    # one Adam step, diffusion.py:211/236
    R.adam_step(params, grads, {})
    for k, pv in params.items():
        if is_pre_bn_bias(k):
            # |g| ~ 1e-9 is rounding noise of the order of Adam's eps: the
            # reference's own update lr*g/(|g|+eps) of these is noise too
            continue
        kk = k.replace(".", "__")
        head = torch.from_numpy(d[f"adam_head__{kk}"])
        got = pv.contiguous().view(-1)[: head.numel()]
        # the first Adam step moves a weight by lr*g/(|g|+1e-8): saturated at
        # +-lr where |g| >> eps, sensitive to rounding of g where it is not
        ghead = torch.from_numpy(d[f"ghead__{kk}"]).abs()
        tol = torch.where(ghead > 1e-5, torch.tensor(2e-6), torch.tensor(1.05e-3))
        assert bool(((got - head).abs() <= tol).all()), k


@pytest.mark.parametrize("name,cond", [("sample_T20_n4_uncond", False), ("sample_T20_n4_cond", True)])
def test_sample_chain_T20(golden_dir, name, cond):
    d = load(golden_dir, name)
    sd = make_state_dict(int(d["seed"]), cond)
    T = int(d["T"])
    sched = R.Schedule(num_timesteps=T)
    y = torch.from_numpy(d["y"]) if cond else None
    zs = torch.from_numpy(d["zs"])
    rec = {19: None, 10: None, 1: None, 0: None}
    x = R.sample_chain(sd, sched, torch.from_numpy(d["x_T"]), zs, y, rec)
    for t, v in rec.items():
        assert rel_mse(v, torch.from_numpy(d[f"x_after_t{t}"])) < 1e-9, t
    assert rel_mse(x, torch.from_numpy(d["final"])) < 1e-9


def test_sample_rng_order(golden_dir):
    """sample() draws x_T first, then one randn_like per step for t = T-1..1
    and nothing at t = 0 (diffusion.py:257, 267-270)."""
    d = load(golden_dir, "sample_T1000_n4_uncond")
    torch.manual_seed(int(d["rng_seed"]))
    x_T = torch.randn(4, 1, 28, 28)
    assert np.array_equal(x_T.numpy(), d["x_T"])
    T = int(d["T"])
    zs = {}
    for t in reversed(range(T)):
        if t > 0:
            zs[t] = torch.randn_like(x_T)
    assert np.array_equal(torch.stack([zs[T - 3], zs[T - 2], zs[T - 1]]).numpy(), d["zs_head"])
    assert np.array_equal(torch.stack([zs[1], zs[2], zs[3]]).numpy(), d["zs_tail"])


@pytest.mark.slow
def test_sample_chain_T1000(golden_dir):
    d = load(golden_dir, "sample_T1000_n4_uncond")
    sd = make_state_dict(int(d["seed"]), False)
    sched = R.Schedule()
    torch.manual_seed(int(d["rng_seed"]))
    x_T = torch.randn(4, 1, 28, 28)
    zs = [None] * 1000
    for t in reversed(range(1000)):
        if t > 0:
            zs[t] = torch.randn_like(x_T)
    rec = {k: None for k in (999, 750, 500, 250, 1, 0)}
    x = R.sample_chain(sd, sched, x_T, zs, None, rec)
    for t, v in rec.items():
        assert rel_mse(v, torch.from_numpy(d[f"x_after_t{t}"])) < 1e-6, t
    assert rel_mse(x, torch.from_numpy(d["final"])) < 1e-6
