"""GPU: the reference's WHOLE `hmc()` run (main_sampling.py:660-774), replayed on its own noise tape.

G14 (oracle/gen_golden.py g14) is the reference's `hmc()` called with the tiny score wrapped to evaluate in float64
(oracle.tiny_score.F64Score: the model is an argument, the reference code is unchanged) on inpaint / sr4 /
deblur_aniso at 32 x 32: the returned [20,3,32,32] tensor, every accept uniform and every -dH.  Here the momentum
draws are regenerated from the run's seed (same torch CPU generator; the stored first and last draws and all uniforms
pin the tape), fed through `sampler.hmc_chains` / `sampler.hmc` on the MI355X, and compared with what the reference
returned: same accept decisions, same 20 reconstructed images within north_star's 1e-4.

Accept-ambiguity band: H is a sum of ~1e3..1e4 in fp32 whose last bits differ between the reference's fp32 torch sums
and the kernels' fp64 partials, so a decision whose uniform lies within BAND of the accept probability is not
determined by the algorithm.  Those iterations (reported) are replayed with the uniform moved to 0 / 1, i.e. with the
reference's own decision; every other decision must come out of the GPU's own energies."""
import copy
import types

import numpy as np
import pytest
import torch

from oracle import operators as oops, schedule as osched
from oracle.tiny_score import F64Score

pytestmark = pytest.mark.gpu
SEQ, SEQ_NEXT = [250, 500, 750], [-1, 250, 500]
T = torch.from_numpy
BAND = 0.02


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def tape_of(g):
    """The draws the reference made: torch.manual_seed(seed); per iteration randn_like(x) then rand(1)."""
    n = len(g['u'])
    torch.manual_seed(int(g['seed']))
    P, U = [], []
    for _ in range(n):
        P.append(torch.randn(1, 3, 32, 32))
        U.append(float(torch.rand(1)))
    assert np.array_equal(np.array(U), g['u']) and np.array_equal(P[0].numpy(), g['p0']) and np.array_equal(P[-1].numpy(), g['p_last'])
    return P


@pytest.mark.parametrize('deg', ['inpaint', 'sr4', 'aniso'])
def test_whole_reference_run_on_the_gpu(golden, tiny_score, deg):
    import nhmc.operators as ops
    from nhmc import plugin, sampler
    g = golden(f'g14_hmc_f64_{deg}_32.npz')
    dev = torch.device('cuda')
    if deg == 'inpaint':
        op = ops.Inpainting(3, 32, T(g['missing']), dev)
    elif deg == 'sr4':
        op = ops.SuperResolution(3, 32, 4, dev)
    else:
        op = ops.Deblurring2D.from_factors(*(T(g[f'aniso_{k}']) for k in ('U1', 'U2', 'V1', 'V2', 'D')), dev)
    P = tape_of(g)
    prob = np.minimum(1.0, np.exp(np.minimum(g['neg_dH'], 50.0)))
    ref_acc = g['u'] < prob
    assert int(ref_acc.sum()) == 100                                              # the reference accepted 100 epochs
    ambiguous = np.abs(g['u'] - prob) < BAND
    u_play = np.where(ambiguous, np.where(ref_acc, 0.0, 1.0), g['u']).astype(np.float32)
    algo = plugin.HMC(F64Score(tiny_score).to(dev), op, float(g['sigma_0']))
    opt = types.SimpleNamespace(tau=float(g['tau']), epsilon=float(g['epsilon']), m=float(g['m']), sigma_0=float(g['sigma_0']), quiet=True)
    noise = sampler.TapeNoise(lambda it: P[it], lambda it: torch.tensor([u_play[it]]))
    b = osched.betas_fp32().to(dev)
    res = sampler.hmc_chains(T(g['x']).to(dev), b, SEQ, SEQ_NEXT, algo, opt, T(g['y_0']).to(dev), op, T(g['x_orig']).to(dev),
                             noise=noise, collect_trace=True)
    assert res.iters == len(g['u'])
    got_acc = np.array([bool(r['accept'][0]) for r in res.trace])
    got_dH = np.array([float(r['dH'][0]) for r in res.trace])
    assert np.array_equal(got_acc, ref_acc), np.nonzero(got_acc != ref_acc)[0][:5]
    # the energies themselves: dH against the reference's, where it is not astronomically large
    small = np.abs(g['neg_dH']) < 50
    assert np.max(np.abs(got_dH[small] + g['neg_dH'][small])) < 0.05, np.max(np.abs(got_dH[small] + g['neg_dH'][small]))
    assert res.samples.shape == (1, 20, 3, 32, 32)
    err = rel(res.samples[0], T(g['out']))
    print(f'{deg}: {len(g["u"])} trajectories, {int(ambiguous.sum())} inside the accept band, returned images rel err {err:.2e}')
    assert err < 1e-4
    # the reference entry point itself returns that tensor
    opt2 = types.SimpleNamespace(**vars(opt), noise_source=sampler.TapeNoise(lambda it: P[it], lambda it: torch.tensor([u_play[it]])))
    out = sampler.hmc(T(g['x']).to(dev), 1, b, SEQ, SEQ_NEXT, algo, opt2, T(g['y_0']).to(dev), op, T(g['x_orig']).to(dev))
    assert out.shape == (20, 3, 32, 32) and torch.equal(out, res.samples[0])
