"""Generate tests/golden/g11_hmc_mass_16.npz by RUNNING THE REFERENCE's `hmc_test_conditioning`
(main_sampling.py:776-894; build container only; same rules as oracle/gen_golden.py).

    python oracle/gen_golden_mass.py            # G11: the reference as it is (default, unstable torch.sort at :860)
    python oracle/gen_golden_mass.py stable     # G11b: g11b_hmc_mass_stable_16.npz
    python oracle/gen_golden_mass.py stable 64  # G11c: g11c_hmc_mass_stable_64.npz (12 288 elements per rank transform)

G11b pins the tie rule the product implements.  The rank transform sorts the per-element variance with torch's default
sort (:860); among EQUAL variances (all of them are zero for the accepts of epochs 14..18, before the Welford
accumulation has started) the ranks -- hence the mass -- are whatever that sort implementation does with ties, which
no other implementation can be asked to reproduce (torch's CPU and GPU sorts disagree with each other).  For G11b the
reference function is run with `torch.sort` wrapped IN THIS SCRIPT so that a call without `stable=` is made with
`stable=True` (ties by index, what the build's radix sort does); main_sampling.py is untouched and nothing else the
reference computes changes.  As for G14, the tiny score is evaluated in float64 (the model is an argument).  Also
recorded, because they are the reference HOST's libm and not arithmetic a kernel could be asked to match: the mass
tables by rank, M = exp(2 rank/(N-1) - 1) and sqrt(M), as `torch.exp` / `torch.sqrt` returned them at :866-867."""
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


def main(stable=False, dim=16):
    ms = import_reference()
    from algos.unconditional import Unconditional
    ops, missing = build_ops(ms, dim, seed=1100)
    Hf = ops['inpaint']
    net = tiny_model()
    if stable:
        from oracle.tiny_score import F64Score
        net = F64Score(net)
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
    rec = dict(u=[], neg_dH=[], p=[], tables={}, sorts=0, ties=0)
    real_rand, real_exp, real_sort, real_sqrt, real_randn_like = torch.rand, torch.exp, torch.sort, torch.sqrt, torch.randn_like
    n_elem = x.numel()

    def rand(*a, **k):
        out = real_rand(*a, **k)
        rec['u'].append(float(out.reshape(-1)[0]))
        return out

    def exp(t, *a, **k):
        if t.numel() == 1 and t.dim() == 1:
            rec['neg_dH'].append(float(t.detach().reshape(-1)[0]))
        return real_exp(t, *a, **k)

    def randn_like(*a, **k):
        out = real_randn_like(*a, **k)
        rec['p'] = rec['p'][:1] + [out.clone()]                 # first and last draw
        return out

    def sort(t, *a, **k):
        if stable and 'stable' not in k and t.numel() == n_elem:
            rec['sorts'] += 1
            rec['ties'] += int(t.numel() - torch.unique(t).numel())
            return real_sort(t, *a, stable=True, **k)
        return real_sort(t, *a, **k)

    def sqrt(t, *a, **k):
        out = real_sqrt(t, *a, **k)
        if torch.is_tensor(t) and t.numel() == n_elem and t.dim() == 1 and float(t.min()) != float(t.max()):
            order = real_sort(t)[1]                              # M is strictly increasing in the rank
            rec['tables'] = dict(M=t[order].clone(), std=out[order].clone())
        return out

    torch.manual_seed(5678)
    torch.rand, torch.exp, torch.sort, torch.sqrt, torch.randn_like = rand, exp, sort, sqrt, randn_like
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            out = ms.hmc_test_conditioning(x, 1, b, [250, 500, 750], [-1, 250, 500], algo, opt, y_0, Hf, x_orig)
    finally:
        torch.rand, torch.exp, torch.sort, torch.sqrt, torch.randn_like = real_rand, real_exp, real_sort, real_sqrt, real_randn_like
    if stable:
        assert rec['sorts'] > 0 and len(rec['tables']) == 2
        M = rec['tables']['M']
        assert bool((M[1:] > M[:-1]).all()), 'the mass table is not strictly increasing in the rank'
        save('g11b_hmc_mass_stable_16.npz' if dim == 16 else f'g11c_hmc_mass_stable_{dim}.npz', x=np32(x), y_0=np32(y_0), x_orig=np32(x_orig), missing=np32(missing),
             seed=np.array(5678), sigma_0=np.array(sigma_0), tau=np.array(0.4), epsilon=np.array(0.05), out=np32(out),
             u=np.array(rec['u']), neg_dH=np.array(rec['neg_dH']), p0=np32(rec['p'][0]), p_last=np32(rec['p'][-1]),
             M_by_rank=np32(M), std_by_rank=np32(rec['tables']['std']), sorts=np.array(rec['sorts']),
             tied_elements=np.array(rec['ties']),
             note=np.array('reference hmc_test_conditioning with torch.sort(stable=True) supplied by the generator wrapper '
                           '(ties by index) and the float64 tiny score; nothing else the reference computes is changed'))
        print('iterations', len(rec['u']), 'returned', tuple(out.shape), 'rank transforms', rec['sorts'], 'tied elements', rec['ties'])
        return
    assert dim == 16
    save('g11_hmc_mass_16.npz', x=np32(x), y_0=np32(y_0), x_orig=np32(x_orig), missing=np32(missing), seed=np.array(5678),
         sigma_0=np.array(sigma_0), tau=np.array(0.4), epsilon=np.array(0.05), out=np32(out), u=np.array(rec['u']),
         neg_dH=np.array(rec['neg_dH']))
    print('iterations', len(rec['u']), 'returned', tuple(out.shape))


if __name__ == '__main__':
    main(stable=len(sys.argv) > 1 and sys.argv[1] == 'stable', dim=int(sys.argv[2]) if len(sys.argv) > 2 else 16)
