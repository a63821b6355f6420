"""CPU: the LAION-shaped oracle (oracle/ref_laion.py) against vectors produced by the
reference's own conditional_diffusion_laion.NoiseModel (tools/make_golden.py)."""
import os

import numpy as np
import torch

from oracle import ref_cpu as R
from oracle import ref_laion as RL
from oracle.weights import key_shapes_laion, make_state_dict_laion

torch.set_num_threads(min(8, os.cpu_count() or 1))


def rel_mse(a, b):
    a, b = a.double(), b.double()
    return ((a - b) ** 2).mean().item() / max((b**2).mean().item(), 1e-30)


def is_pre_bn_bias(key):
    stage, idx, kind = (key.split(".") + ["", ""])[:3]
    return kind == "bias" and idx in ("0", "3") and stage[:3] in ("enc", "dec", "bot")


def test_laion_layout():
    ks = key_shapes_laion()
    n = sum(int(np.prod(s)) for k, s, kind in ks if not kind.startswith("bn_r") and kind != "bn_nbt")
    assert n == 5_793_124


def test_laion_oracle_matches_reference(golden_dir):
    d = np.load(os.path.join(golden_dir, "laion_B8.npz"))
    sd = make_state_dict_laion(0)
    x_t, t = torch.from_numpy(d["x_t"]), torch.from_numpy(d["t"])
    noise, cond = torch.from_numpy(d["noise"]), torch.from_numpy(d["cond"])
    assert np.array_equal(RL.timestep_embedding(torch.tensor([0, 1, 999]), 768).numpy(), d["sinusoid"])
    for mode in ("train", "eval"):
        p, b = R.split_state(sd)
        with torch.no_grad():
            eps = RL.unet_forward(p, b, x_t, t, cond, training=(mode == "train"))
        assert rel_mse(eps, torch.from_numpy(d[f"eps_{mode}"])) < 1e-10, mode
        if mode == "train":
            for k, v in b.items():
                if v.dtype != torch.int64:
                    assert torch.allclose(v, torch.from_numpy(d["buf__" + k.replace(".", "__")]), rtol=1e-5, atol=1e-5), k
    loss, eps, grads, _ = RL.train_step_grads(sd, x_t, t, noise, cond)
    assert abs(loss.item() - float(d["loss_train"])) < 1e-5 * float(d["loss_train"])
    for k, g in grads.items():
        kk = k.replace(".", "__")
        gn = float(d[f"gnorm__{kk}"])
        got = g.contiguous().view(-1)
        if is_pre_bn_bias(k):
            wn = float(d[f"gnorm__{kk.replace('bias', 'weight')}"])
            assert gn <= 1e-5 * wn and got.double().norm().item() <= 1e-5 * wn, k
            continue
        assert abs(got.double().norm().item() - gn) <= 1e-3 * gn, k
    # reverse chain with recorded noise
    T = int(d["chain_T"])
    x = RL.sample_chain(sd, R.Schedule(num_timesteps=T), torch.from_numpy(d["chain_x_T"]),
                        torch.from_numpy(d["chain_zs"]), cond[:2])
    assert rel_mse(x, torch.from_numpy(d["chain_final"])) < 1e-9


def test_laion_oracle_matches_reference_at_64x64(golden_dir):
    """BASELINE.json configs[4] resolution: the reference class is fully convolutional
    (conditional_diffusion_laion.py:304-332) and so is the oracle; golden from the reference itself."""
    d = np.load(os.path.join(golden_dir, "laion_B2_hw64.npz"))
    sd = make_state_dict_laion(0)
    x_t, t = torch.from_numpy(d["x_t"]), torch.from_numpy(d["t"])
    noise, cond = torch.from_numpy(d["noise"]), torch.from_numpy(d["cond"])
    assert x_t.shape == (2, 4, 64, 64)
    for mode in ("train", "eval"):
        p, b = R.split_state(sd)
        with torch.no_grad():
            eps = RL.unet_forward(p, b, x_t, t, cond, training=(mode == "train"))
        assert rel_mse(eps, torch.from_numpy(d[f"eps_{mode}"])) < 1e-10, mode
        if mode == "train":
            for k, v in b.items():
                if v.dtype != torch.int64:
                    assert torch.allclose(v, torch.from_numpy(d["buf__" + k.replace(".", "__")]), rtol=1e-5, atol=1e-5), k
    loss, eps, grads, _ = RL.train_step_grads(sd, x_t, t, noise, cond)
    assert abs(loss.item() - float(d["loss_train"])) < 1e-5 * float(d["loss_train"])
    for k, g in grads.items():
        if is_pre_bn_bias(k):
            continue
        gn = float(d["gnorm__" + k.replace(".", "__")])
        assert abs(g.double().norm().item() - gn) <= 1e-3 * gn, k
    T = int(d["chain_T"])
    x = RL.sample_chain(sd, R.Schedule(num_timesteps=T), torch.from_numpy(d["chain_x_T"]),
                        torch.from_numpy(d["chain_zs"]), cond[:1])
    assert rel_mse(x, torch.from_numpy(d["chain_final"])) < 1e-9
