"""CPU oracle for the tiny-diffusion DDPM hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  The product package
(``tiny_diffusion_amd``) never imports from here and fails loudly when its HIP
library is missing.

Parity status: PINNED.  ``tests/golden/*.npz`` were produced by running the
reference's own ``NoiseModel`` / ``ForwardProcess`` / ``sample`` code
(/root/reference/diffusion.py, conditional_diffusion.py) in the build
container with ``tools/make_golden.py``; ``tests/test_oracle_golden.py`` checks
this restatement against every one of those vectors.
"""
