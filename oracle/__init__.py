"""CPU oracle for the noise-space HMC hot path (TEST INFRASTRUCTURE ONLY).

This package is a CPU restatement, in plain PyTorch-CPU / numpy, of the
reference's algorithm for the path named in BASELINE.json (`main_sampling.py`
`hmc`, `iterative_sampling`, `compute_alpha`; `algos/unconditional.py`;
`obs_functions/Hfuncs.py` Inpainting / SuperResolution / Deblurring2D).

It is the *checker*, never the product:
  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
    leg may import it;
  * nothing under `noise-space-hmc_amd/` imports it, and the product path
    raises if the HIP library is missing instead of falling back to this.

Pinning: the reference ships no tests, golden vectors or fixtures for this
path, so the oracle is pinned against outputs of the
reference itself, produced in the build container by `oracle/gen_golden.py`
(which imports `/root/reference`) and committed as data under `tests/golden/`.
`tests/test_oracle_golden.py` replays every fixture through this package.
"""

from . import schedule, operators, ddim, hmc_ref, latent_ref, philox_ref, tiny_score  # noqa: F401
