"""Generate tests/golden/g7_hmc_latent_16.npz by RUNNING THE REFERENCE's `hmc_latent` (build container only).

Same rules as oracle/gen_golden.py: imports /root/reference read-only, inert placeholder modules for the absent
off-path packages (torchvision, skimage, lpips, omegaconf), data only into the repo.  The LatentDiffusion class
itself needs pytorch_lightning + taming (absent), so the model handed to `hmc_latent` is the duck-typed
`oracle.latent_ref.TinyLatentModel` -- `hmc_latent` only calls apply_model / differentiable_decode_first_stage /
alphas_cumprod(_prev) on it (main_sampling_latent.py:651,670,771-772; algos/unconditional_latent.py:12).
"""
import argparse
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.gen_golden import import_reference, np32, save, REF  # noqa: E402


def main():
    import_reference()                                           # placeholders + sys.path
    om = types.ModuleType('omegaconf')
    om.OmegaConf = None
    sys.modules['omegaconf'] = om
    import main_sampling_latent as msl
    msl.device = torch.device('cpu')
    msl.config = msl.dict2namespace({'data': {'rescaled': True, 'logit_transform': False}})
    from obs_functions.Hfuncs import Inpainting
    from algos.unconditional_latent import Unconditional_Latent
    from oracle.latent_ref import TinyLatentModel

    dim, zdim = 64, 16
    g = torch.Generator().manual_seed(1700)
    r = 3 * torch.randperm(dim * dim, generator=g)[: int(dim * dim * 0.92)].long()
    missing = torch.cat([r, r + 1, r + 2])
    Hf = Inpainting(3, dim, missing, 'cpu')
    model = TinyLatentModel()
    x_orig = torch.rand(1, 3, dim, dim, generator=g) * 2 - 1
    sigma_0 = 2 * 0.05
    y_0 = Hf.H(x_orig) + sigma_0 * torch.randn(1, Hf.kept_indices.numel(), generator=g)
    x = torch.randn(1, 3, zdim, zdim, generator=g)
    opt = argparse.Namespace(tau=0.3, epsilon=0.1, m=1.0, sigma_0=sigma_0, sigma_y=0.5, algo='hmc_latent', noise='ddpm',
                             image_folder='/tmp/nhmc_golden_scratch')
    algo = Unconditional_Latent(model, Hf, sigma_0)
    rec = dict(p=[], u=[], neg_dH=[])
    real_randn_like, real_rand, real_exp = torch.randn_like, torch.rand, torch.exp

    def randn_like(*a, **k):
        out = real_randn_like(*a, **k)
        if not rec['p']:
            rec['p'].append(out.clone())
        return out

    def rand(*a, **k):
        out = real_rand(*a, **k)
        rec['u'].append(float(out.reshape(-1)[0]))
        return out

    def exp(t, *a, **k):
        if t.numel() == 1 and t.dim() == 1:
            rec['neg_dH'].append(float(t.detach().reshape(-1)[0]))
        return real_exp(t, *a, **k)

    torch.manual_seed(5678)
    torch.randn_like, torch.rand, torch.exp = randn_like, rand, exp
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            out = msl.hmc_latent(x, 1, [250, 500, 750], [-1, 250, 500], algo, opt, y_0, Hf, x_orig)
    finally:
        torch.randn_like, torch.rand, torch.exp = real_randn_like, real_rand, real_exp
    save('g7_hmc_latent_16.npz', x=np32(x), y_0=np32(y_0), x_orig=np32(x_orig), missing=np32(missing), seed=np.array(5678),
         sigma_0=np.array(sigma_0), sigma_y=np.array(0.5), tau=np.array(0.3), epsilon=np.array(0.1), m=np.array(1.0),
         out=np32(out), u=np.array(rec['u']), neg_dH=np.array(rec['neg_dH']), p0=np32(rec['p'][0]))
    print('iterations', len(rec['u']), 'returned', tuple(out.shape))


if __name__ == '__main__':
    main()
