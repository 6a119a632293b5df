"""Generate tests/golden/g11_hmc_mass_16.npz by RUNNING THE REFERENCE's `hmc_test_conditioning`
(main_sampling.py:776-894; build container only; same rules as oracle/gen_golden.py)."""
import argparse
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.gen_golden import import_reference, np32, save, build_ops, tiny_model  # noqa: E402


def main():
    ms = import_reference()
    from algos.unconditional import Unconditional
    dim = 16
    ops, missing = build_ops(ms, dim, seed=1100)
    Hf = ops['inpaint']
    net = tiny_model()
    b = torch.from_numpy(ms.get_beta_schedule(beta_schedule='linear', beta_start=1e-4, beta_end=0.02,
                                              num_diffusion_timesteps=1000)).float()
    g = torch.Generator().manual_seed(12)
    x_orig = torch.rand(1, 3, dim, dim, generator=g) * 2 - 1
    sigma_0 = 0.1
    y_0 = Hf.H(x_orig) + sigma_0 * torch.randn(1, Hf.kept_indices.numel(), generator=g)
    x = torch.randn(1, 3, dim, dim, generator=g)
    opt = argparse.Namespace(tau=0.4, epsilon=0.05, m=1.0, sigma_0=sigma_0, algo='hmc_mass', noise='ddpm',
                             image_folder='/tmp/nhmc_golden_scratch')
    os.makedirs(opt.image_folder, exist_ok=True)
    algo = Unconditional(net, Hf, sigma_0)
    rec = dict(u=[], neg_dH=[])
    real_rand, real_exp = torch.rand, torch.exp

    def rand(*a, **k):
        out = real_rand(*a, **k)
        rec['u'].append(float(out.reshape(-1)[0]))
        return out

    def exp(t, *a, **k):
        if t.numel() == 1 and t.dim() == 1:
            rec['neg_dH'].append(float(t.detach().reshape(-1)[0]))
        return real_exp(t, *a, **k)

    torch.manual_seed(5678)
    torch.rand, torch.exp = rand, exp
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            out = ms.hmc_test_conditioning(x, 1, b, [250, 500, 750], [-1, 250, 500], algo, opt, y_0, Hf, x_orig)
    finally:
        torch.rand, torch.exp = real_rand, real_exp
    save('g11_hmc_mass_16.npz', x=np32(x), y_0=np32(y_0), x_orig=np32(x_orig), missing=np32(missing), seed=np.array(5678),
         sigma_0=np.array(sigma_0), tau=np.array(0.4), epsilon=np.array(0.05), out=np32(out), u=np.array(rec['u']),
         neg_dH=np.array(rec['neg_dH']))
    print('iterations', len(rec['u']), 'returned', tuple(out.shape))


if __name__ == '__main__':
    main()
