"""GPU: the latent-space sampler (hmc_latent path) against the latent oracle and the reference-captured G7 run."""
import copy
import types

import numpy as np
import pytest
import torch

from oracle import latent_ref, operators as oops

pytestmark = pytest.mark.gpu
T = torch.from_numpy
SEQ, SEQ_NEXT = [250, 500, 750], [-1, 250, 500]


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


class F64Latent(torch.nn.Module):
    """TinyLatentModel evaluated in fp64 and rounded to fp32 (see F64Score in test_sampler_gpu.py): removes the
    CPU-vs-GPU convolution noise that a 70-trajectory comparison would otherwise amplify."""

    def __init__(self):
        super().__init__()
        self.m = latent_ref.TinyLatentModel().double()
        self.alphas_cumprod = self.m.alphas_cumprod.float()
        self.alphas_cumprod_prev = self.m.alphas_cumprod_prev.float()

    def to(self, dev):
        self.m = self.m.to(dev)
        self.alphas_cumprod, self.alphas_cumprod_prev = self.alphas_cumprod.to(dev), self.alphas_cumprod_prev.to(dev)
        return self

    def apply_model(self, x, t, cond=None):
        return self.m.apply_model(x.double(), t.double(), cond).float()

    def differentiable_decode_first_stage(self, z):
        return self.m.differentiable_decode_first_stage(z.double()).float()


def test_first_latent_trajectory_of_the_reference_run(golden):
    from nhmc import operators, plugin, sampler
    g = golden('g7_hmc_latent_16.npz')
    dev = torch.device('cuda')
    op = operators.Inpainting(3, 64, T(g['missing']), dev)
    model = latent_ref.TinyLatentModel().to(dev)
    algo = plugin.HMCLatent(model, op, float(g['sigma_0']))
    table = torch.cat([model.alphas_cumprod_prev[0:1], model.alphas_cumprod])
    eng = sampler.LeapfrogEngine(algo.score, op, None, SEQ, SEQ_NEXT, dev, alpha_table=table,
                                 image_map=model.differentiable_decode_first_stage)
    st = sampler.ChainState(1, 0.3, 0.1, dev)
    st['eps_eff'].fill_(float(g['epsilon']))
    st['sigma_y'].fill_(float(g['sigma_y']))
    got = sampler.run_trajectory(eng, T(g['x']).to(dev), T(g['p0']).to(dev).clone(), T(g['y_0']).to(dev), st, 1.0, 2)
    want = latent_ref.trajectory_latent(T(g['x']), T(g['p0']), SEQ, SEQ_NEXT, latent_ref.TinyLatentModel(),
                                        oops.InpaintRef(3, 64, T(g['missing'])), T(g['y_0']),
                                        sigma_y=float(g['sigma_y']), eps=float(g['epsilon']), m=1.0, L=2)
    assert rel(got['x_prop'], want['x']) < 1e-4 and rel(got['xt'], want['xt']) < 1e-4 and rel(got['loss'], want['loss']) < 1e-4
    assert abs(float((got['H1'] - got['H0'])[0]) + float(g['neg_dH'][0])) < 0.02


def test_latent_loop_takes_the_oracles_decisions():
    """70-epoch latent loop at B = 1 on the noise the oracle draws: same accept decisions, same returned latents."""
    from nhmc import operators, plugin, sampler
    dev = torch.device('cuda')
    g_ = torch.Generator().manual_seed(4)
    missing = oops.random_inpaint_missing(64, generator=g_)
    ref_op, op = oops.InpaintRef(3, 64, missing), operators.Inpainting(3, 64, missing, dev)
    x = torch.randn(1, 3, 16, 16, generator=g_)
    x_orig = torch.rand(1, 3, 64, 64, generator=g_) * 2 - 1
    y = ref_op.H(x_orig) + 0.1 * torch.randn(1, ref_op.M, generator=g_)
    kw = dict(sigma_y=0.5, tau=0.3, epsilon=0.1, m=1.0, sigma_0=0.1)
    cpu_model = F64Latent()                                 # built BEFORE seeding: nn layer init draws from the global RNG
    torch.manual_seed(99)
    trace = {}
    want = latent_ref.hmc_latent_reference(x, SEQ, SEQ_NEXT, cpu_model, ref_op, y, x_orig, trace=trace, **kw)
    torch.manual_seed(99)                                   # regenerate the very same draws as a tape
    P, U = [], []
    for _ in range(70):
        P.append(torch.randn(1, 3, 16, 16))
        U.append(torch.rand(1))
    algo = plugin.HMCLatent(F64Latent().to(dev), op, 0.1)
    opt = types.SimpleNamespace(tau=0.3, epsilon=0.1, m=1.0, sigma_0=0.1, sigma_y=0.5)
    res = sampler.hmc_latent_chains(x.to(dev), SEQ, SEQ_NEXT, algo, opt, y.to(dev), op, x_orig.to(dev),
                                    noise=sampler.TapeNoise(lambda it: P[it], lambda it: U[it]), collect_trace=True)
    got_acc = [bool(r['accept'][0]) for r in res.trace]
    for it, (a, b) in enumerate(zip(trace['accept'], got_acc)):
        margin = abs(float(U[it]) - min(1.0, float(np.exp(-trace['dH'][it]))))
        assert a == b or margin < 1e-3, (it, a, b, margin)
    assert [float(r['sigma_y'][0]) for r in res.trace] == trace['sigma_y']
    assert res.samples[0].shape == want.shape and rel(res.samples[0], want) < 1e-4
    out = sampler.hmc_latent(x.to(dev), 1, SEQ, SEQ_NEXT, algo,
                             types.SimpleNamespace(**vars(opt), noise_source=sampler.TapeNoise(lambda it: P[it], lambda it: U[it])),
                             y.to(dev), op, x_orig.to(dev))
    assert torch.equal(out, res.samples[0])
