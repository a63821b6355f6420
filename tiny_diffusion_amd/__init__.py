"""tiny_diffusion_amd - the DDPM hot path of david-wb/tiny-diffusion on MI355X.

    from tiny_diffusion_amd.diffusion import NoiseModel, ForwardProcess, sample
    from tiny_diffusion_amd.conditional_diffusion import NoiseModel, ForwardProcess, sample
    from tiny_diffusion_amd.conditional_diffusion_laion import NoiseModel, ForwardProcess, sample
    from tiny_diffusion_amd.latent_diffusion import NoiseModel, ForwardProcess, sample, VAE, VAEConfig

Importing the package loads libtdx.so (built in-tree by ``python -m
tiny_diffusion_amd._build``); there is no CPU fallback.
"""
from . import _lib  # noqa: F401  (fails loudly if the HIP library is missing)
from . import diffusion, conditional_diffusion, conditional_diffusion_laion, latent_diffusion, vae, diffusion_transformer  # noqa: F401
from .schedule import ForwardProcess, p_sample_step, sample_loop  # noqa: F401

__version__ = "0.1.0"
