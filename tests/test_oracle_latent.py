"""CPU: the latent-space oracle (oracle/ref_latent.py) against vectors produced by the
reference's own latent_diffusion.NoiseModel and vae.VAE (tools/make_golden.py)."""
import os

import numpy as np
import torch

from oracle import ref_cpu as R
from oracle import ref_latent as RL
from oracle.weights import (key_shapes_latent, make_state_dict_latent, make_state_dict_vae, state_dict_sha256)

torch.set_num_threads(min(8, os.cpu_count() or 1))


def rel_mse(a, b):
    a, b = a.double(), b.double()
    return ((a - b) ** 2).mean().item() / max((b**2).mean().item(), 1e-30)


def is_pre_bn_bias(key):
    """Parameters whose true gradient is exactly zero under train-mode BatchNorm: the Linear
    biases in front of a BN, and every bias that shifts the time signal by the same vector for
    all samples (time_embedding.2.bias, time_proj*.bias) - the decoder adds that signal in
    front of Linear+BN (latent_diffusion.py:124-126), which removes batch-constant shifts."""
    stage, idx, kind = (key.split(".") + ["", ""])[:3]
    if key in ("time_embedding.2.bias", "time_proj1.bias", "time_proj2.bias", "time_proj3.bias", "initial_fc.bias"):
        return True
    return kind == "bias" and idx in ("0", "3") and stage[:3] in ("enc", "dec", "bot")


def test_latent_layout_and_weight_identity(golden_dir):
    ks = key_shapes_latent()
    n = sum(int(np.prod(s)) for k, s, kind in ks if not kind.startswith("bn_r") and kind != "bn_nbt")
    assert n == 1_422_932  # sum(p.numel()) of the reference model (golden generation asserts strict load)
    lines = dict(l.rsplit(" ", 1) for l in open(os.path.join(golden_dir, "weights.sha256")).read().splitlines())
    assert state_dict_sha256(make_state_dict_latent(0)) == lines["latent seed0"]
    assert state_dict_sha256(make_state_dict_vae(0)) == lines["vae seed0"]


def test_latent_oracle_matches_reference(golden_dir):
    d = np.load(os.path.join(golden_dir, "latent_B32.npz"))
    sd, vp = make_state_dict_latent(0), make_state_dict_vae(0)
    # VAE encode / reparameterize / decode (vae.py:51-62)
    mu, logvar = RL.vae_encode(vp, torch.from_numpy(d["x_img"]))
    assert torch.allclose(mu, torch.from_numpy(d["mu"]), rtol=1e-5, atol=1e-6)
    assert torch.allclose(logvar, torch.from_numpy(d["logvar"]), rtol=1e-5, atol=1e-6)
    z0 = RL.vae_reparameterize(mu, logvar, torch.from_numpy(d["eps_vae"]))
    assert torch.allclose(z0, torch.from_numpy(d["z0"]), rtol=1e-5, atol=1e-6)
    assert torch.allclose(RL.vae_decode(vp, torch.from_numpy(d["z0"])), torch.from_numpy(d["dec"]), rtol=1e-5, atol=1e-6)
    # 2-D q_sample (latent_diffusion.py:143-154)
    z_t, t = torch.from_numpy(d["z_t"]), torch.from_numpy(d["t"])
    noise, y = torch.from_numpy(d["noise"]), torch.from_numpy(d["y"])
    zt = RL.q_sample(R.Schedule(), torch.from_numpy(d["z0"]), t, noise)
    assert torch.allclose(zt, z_t, rtol=1e-6, atol=1e-6)
    for mode in ("train", "eval"):
        p, b = R.split_state(sd)
        with torch.no_grad():
            eps = RL.noise_forward(p, b, z_t, t, y, training=(mode == "train"))
        assert rel_mse(eps, torch.from_numpy(d[f"eps_{mode}"])) < 1e-10, mode
        if mode == "train":
            for k, v in b.items():
                if v.dtype != torch.int64:
                    assert torch.allclose(v, torch.from_numpy(d["buf__" + k.replace(".", "__")]), rtol=1e-5, atol=1e-5), k
                else:
                    assert int(v) == 2
    loss, eps, grads, _ = RL.train_step_grads(sd, z_t, t, noise, y)
    assert abs(loss.item() - float(d["loss_train"])) < 1e-5 * float(d["loss_train"])
    for k, g in grads.items():
        kk = k.replace(".", "__")
        gn = float(d[f"gnorm__{kk}"])
        got = g.contiguous().view(-1)
        if is_pre_bn_bias(k):
            wn = float(d[f"gnorm__{kk.replace('bias', 'weight')}"])
            assert gn <= 2e-5 * wn and got.double().norm().item() <= 2e-5 * wn, (k, gn, wn)
            continue
        assert abs(got.double().norm().item() - gn) <= 1e-3 * gn, k
        head = torch.from_numpy(d[f"ghead__{kk}"])
        rms = gn / np.sqrt(got.numel())
        assert (got[: head.numel()] - head).abs().max().item() <= 2e-2 * max(head.abs().max().item(), rms), k
    # reverse chain with recorded noise + decode (latent_diffusion.py:321-346)
    T = int(d["chain_T"])
    z, img = RL.sample_chain(sd, vp, R.Schedule(num_timesteps=T), torch.from_numpy(d["chain_z_T"]),
                             torch.from_numpy(d["chain_zs"]), y[:4])
    assert rel_mse(z, torch.from_numpy(d["chain_final"])) < 1e-9
    assert torch.allclose(img, torch.from_numpy(d["chain_img"]), rtol=1e-4, atol=1e-5) and img.shape == (4, 1, 28, 28)
