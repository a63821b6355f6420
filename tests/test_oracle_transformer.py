"""CPU: oracle/ref_transformer.py against vectors produced by the reference's own
diffusion_transformer.NoiseModel (tools/make_golden.py), and the host module's layout / default
initialisation against the reference's."""
import os

import numpy as np
import torch

from oracle import ref_transformer as RT
from oracle.weights import key_shapes_transformer, make_state_dict_transformer, state_dict_sha256

torch.set_num_threads(min(8, os.cpu_count() or 1))


def rel_mse(a, b):
    a, b = a.double(), b.double()
    return ((a - b) ** 2).mean().item() / max((b**2).mean().item(), 1e-30)


def test_transformer_oracle_matches_reference(golden_dir):
    d = np.load(os.path.join(golden_dir, "transformer_B16.npz"))
    sd = make_state_dict_transformer(0)
    lines = dict(l.rsplit(" ", 1) for l in open(os.path.join(golden_dir, "weights.sha256")).read().splitlines())
    assert state_dict_sha256(sd) == lines["transformer seed0"]
    z_t, t, y, noise = (torch.from_numpy(d[k]) for k in ("z_t", "t", "y", "noise"))
    with torch.no_grad():
        eps = RT.noise_forward(sd, z_t, t, y)
    assert rel_mse(eps, torch.from_numpy(d["eps_eval"])) < 1e-10
    assert rel_mse(eps, torch.from_numpy(d["eps_train_nodrop"])) < 1e-10   # no BatchNorm: same in train mode
    loss, _, grads = RT.train_step_grads(sd, z_t, t, noise, y)
    assert abs(loss.item() - float(d["loss"])) < 1e-5 * float(d["loss"])
    D = 256
    for k, g in grads.items():
        kk = k.replace(".", "__")
        gn = float(d[f"gnorm__{kk}"])
        got = g.contiguous().view(-1)
        assert abs(got.double().norm().item() - gn) <= 1e-3 * gn + 1e-12, k
        head = torch.from_numpy(d[f"ghead__{kk}"])
        assert (got[: head.numel()] - head).abs().max().item() <= 2e-2 * max(head.abs().max().item(), gn / got.numel() ** 0.5) + 1e-12, k
        if k.endswith("attention.in_proj_weight"):
            # length-1 sequence: the softmax has one entry, so Q and K receive exactly zero gradient
            assert float(g[: 2 * D].abs().max()) == 0.0


def test_transformer_module_layout_and_init(golden_dir):
    from tiny_diffusion_amd import _lib
    from tiny_diffusion_amd.diffusion_transformer import NoiseModel

    lines = dict(l.rsplit(" ", 1) for l in open(os.path.join(golden_dir, "weights.sha256")).read().splitlines())
    torch.manual_seed(0)
    m = NoiseModel()
    assert state_dict_sha256(m.state_dict()) == lines["init transformer_seed0"]   # the reference's default init
    assert [k for k, _, _ in key_shapes_transformer()] == list(m.state_dict().keys())
    m.load_state_dict(make_state_dict_transformer(0), strict=True)
    import pytest
    with pytest.raises(_lib.TdxError):
        m(torch.zeros(2, 20), torch.zeros(2, dtype=torch.long), torch.zeros(2, dtype=torch.long))
