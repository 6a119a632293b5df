"""GPU: the diagonal-mass sampler (hmc_test_conditioning path) against the oracle."""
import copy
import types

import numpy as np
import pytest
import torch

from oracle import mass_ref, operators as oops, schedule as osched

pytestmark = pytest.mark.gpu
SEQ, SEQ_NEXT = [250, 500, 750], [-1, 250, 500]


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def test_leapfrog_mass_kernel_matches_reference_ops():
    import nhmc.kernels as K
    g_ = torch.Generator().manual_seed(1)
    shape = (2, 3, 32, 32)
    x, z, g, mean0, m20 = (torch.randn(shape, generator=g_) for _ in range(5))
    M = torch.exp(torch.rand(shape, generator=g_) * 2 - 1)
    std, inv = torch.sqrt(M), 1.0 / M
    eps, sig = 0.05, 0.9
    ef, eh, kf = eps, eps / 2, 1 / (2 * sig ** 2)          # python floats: torch rounds them to fp32 per op, as in the reference
    # FIRST
    p = z * std
    Sx, Sp = (x.double() ** 2).sum((1, 2, 3)), (inv * p ** 2).double().sum((1, 2, 3))
    G = x + kf * g
    p1 = p - eh * G
    x1 = x + ef * p1 * inv
    dx, dp = x.cuda(), torch.empty(shape, device='cuda')
    ws = K.leapfrog_ws(2, x[0].numel(), 'cuda')
    K.leapfrog_mass(K.LF_FIRST, dx, dp, g.cuda(), inv.cuda(), eps, sig, ws, z=z.cuda(), std_m=std.cuda())
    assert torch.equal(dx.cpu(), x1) and torch.equal(dp.cpu(), p1)
    tiles = K.leapfrog_tiles(x[0].numel())
    assert rel(K.sum_partials(ws, tiles, 2, 2, 0), Sx) < 1e-6 and rel(K.sum_partials(ws, tiles, 2, 2, 1), Sp) < 1e-6
    # MID with Welford at l = 3 on chain 0 only
    won = torch.tensor([1, 0], dtype=torch.int32)
    G = x1 + kf * g
    p2 = p1 - ef * G
    delta = x1 - mean0
    mean1 = mean0 + delta / 4
    m21 = m20 + delta * (x1 - mean1)
    x2 = x1 + ef * p2 * inv
    dmean, dm2 = mean0.cuda(), m20.cuda()
    K.leapfrog_mass(K.LF_MID, dx, dp, g.cuda(), inv.cuda(), eps, sig, welford_on=won.cuda(), mean=dmean, m2=dm2, l=3)
    assert torch.equal(dx.cpu(), x2) and torch.equal(dp.cpu(), p2)
    assert torch.equal(dmean.cpu()[0], mean1[0]) and torch.equal(dm2.cpu()[0], m21[0])
    assert torch.equal(dmean.cpu()[1], mean0[1]) and torch.equal(dm2.cpu()[1], m20[1])


def test_mass_from_variance_rank_transform():
    import nhmc.kernels as K
    g_ = torch.Generator().manual_seed(2)
    m2 = torch.rand(3, 3, 16, 16, generator=g_)
    inv, std = torch.ones(3, 3, 16, 16, device='cuda'), torch.ones(3, 3, 16, 16, device='cuda')
    flags = torch.tensor([1, 0, 1], dtype=torch.int32).cuda()
    m2[0].view(-1)[100:140] = 0.25                                      # ties: broken by index, as torch.sort(stable=True)
    m2[2].view(-1)[::7] = 0.0
    from nhmc.schedule import mass_tables
    K.mass_from_variance(m2.cuda(), 5, flags, inv, std, mass_tables(m2[0].numel(), 'cuda'))
    for c in range(3):
        M_ref, std_ref, inv_ref = mass_ref.mass_from_variance(m2[c], 5, stable=True)
        if c == 1:
            assert float((inv[c] - 1).abs().max()) == 0 and float((std[c] - 1).abs().max()) == 0
        else:                                                          # the same host tables on both sides: the same bits
            assert torch.equal(inv[c].reshape(-1).cpu(), inv_ref) and torch.equal(std[c].reshape(-1).cpu(), std_ref)


@pytest.mark.parametrize('fixture,dim', [('g11b_hmc_mass_stable_16.npz', 16), ('g11c_hmc_mass_stable_64.npz', 64)])
def test_reference_diagonal_mass_run_on_the_gpu(golden, tiny_score, fixture, dim):
    """G11b = the reference's `hmc_test_conditioning` (main_sampling.py:776-894) run with ties of the rank transform broken
    by index (oracle/gen_golden_mass.py stable; the oracle reproduces it bit for bit, tests/test_mass_cpu.py), replayed on
    the GPU on the run's own tape: every accept decision outside the ambiguity band, the energy differences, and the 35
    returned images within north_star's 1e-4.  The mass tables (sqrt(M_r), 1/M_r by rank) are the recorded ones of the
    host the reference ran on: they are that host's libm, not kernel arithmetic."""
    from nhmc import operators, plugin, sampler
    from oracle.tiny_score import F64Score
    T = torch.from_numpy
    g = golden(fixture)                                                  # G11c: the same reference run at 64 x 64
    dev = torch.device('cuda')
    n = len(g['u'])
    torch.manual_seed(int(g['seed']))
    P, U = [], []
    for _ in range(n):
        P.append(torch.randn(1, 3, dim, dim))
        U.append(float(torch.rand(1)))
    assert np.array_equal(np.array(U), g['u']) and np.array_equal(P[0].numpy(), g['p0']) and np.array_equal(P[-1].numpy(), g['p_last'])
    prob = np.minimum(1.0, np.exp(np.minimum(g['neg_dH'], 50.0)))
    ref_acc = g['u'] < prob
    ambiguous = np.abs(g['u'] - prob) < 0.02
    u_play = np.where(ambiguous, np.where(ref_acc, 0.0, 1.0), g['u']).astype(np.float32)
    op = operators.Inpainting(3, dim, T(g['missing']), dev)
    algo = plugin.HMC(F64Score(tiny_score).to(dev), op, float(g['sigma_0']))
    opt = types.SimpleNamespace(tau=float(g['tau']), epsilon=float(g['epsilon']), sigma_0=float(g['sigma_0']))
    M = T(g['M_by_rank'])
    res = sampler.hmc_mass_chains(T(g['x']).to(dev), osched.betas_fp32().to(dev), SEQ, SEQ_NEXT, algo, opt, T(g['y_0']).to(dev), op,
                                  T(g['x_orig']).to(dev), noise=sampler.TapeNoise(lambda it: P[it], lambda it: torch.tensor([u_play[it]])),
                                  collect_trace=True, max_iters=n, mass_tables=(T(g['std_by_rank']), 1.0 / M))
    assert res.iters == n
    got_acc = np.array([bool(r['accept'][0]) for r in res.trace])
    got_dH = np.array([float(r['dH'][0]) for r in res.trace])
    assert np.array_equal(got_acc, ref_acc), np.nonzero(got_acc != ref_acc)[0][:5]
    small = np.abs(g['neg_dH']) < 50
    worst = float(np.max(np.abs(got_dH[small] + g['neg_dH'][small])))
    err = rel(res.samples[0], T(g['out']))
    print(f'diagonal-mass run: {n} trajectories, {int(ambiguous.sum())} inside the accept band, max |dH - dH_ref| {worst:.4f}, '
          f'returned images rel err {err:.2e} ({int(g["sorts"])} rank transforms, {int(g["tied_elements"])} tied elements)')
    assert worst < 0.05
    assert res.samples.shape == (1, 35, 3, dim, dim) and err < 1e-4


def test_mass_loop_takes_the_oracles_decisions(tiny_score):
    from nhmc import operators, plugin, sampler

    class F64Score(torch.nn.Module):
        def __init__(self, net):
            super().__init__()
            self.net = copy.deepcopy(net).double()

        def forward(self, x, t):
            return self.net(x.double(), t.double()).float()

    dim = 16
    g_ = torch.Generator().manual_seed(3)
    missing = oops.random_inpaint_missing(dim, generator=g_)
    ref_op, op = oops.InpaintRef(3, dim, missing), operators.Inpainting(3, dim, missing, 'cuda')
    x = torch.randn(1, 3, dim, dim, generator=g_)
    x_orig = torch.rand(1, 3, dim, dim, generator=g_) * 2 - 1
    y = ref_op.H(x_orig) + 0.1 * torch.randn(1, ref_op.M, generator=g_)
    cpu_score = F64Score(tiny_score)
    torch.manual_seed(7)
    trace = {}
    want = mass_ref.hmc_mass_reference(x, osched.betas_fp32(), SEQ, SEQ_NEXT, cpu_score, ref_op, y, x_orig, tau=0.2,
                                       epsilon=0.05, sigma_0=0.1, trace=trace, stable_sort=True)   # ties by index, as the radix sort
    torch.manual_seed(7)
    P, U = [], []
    for _ in range(len(trace['accept'])):
        P.append(torch.randn(1, 3, dim, dim))
        U.append(torch.rand(1))
    algo = plugin.HMC(F64Score(tiny_score).cuda(), op, 0.1)
    opt = types.SimpleNamespace(tau=0.2, epsilon=0.05, sigma_0=0.1)
    res = sampler.hmc_mass_chains(x.cuda(), osched.betas_fp32().cuda(), SEQ, SEQ_NEXT, algo, opt, y.cuda(), op, x_orig.cuda(),
                                  noise=sampler.TapeNoise(lambda it: P[it], lambda it: U[it]), collect_trace=True,
                                  max_iters=len(P))
    for it, rec in enumerate(res.trace):
        a, b = trace['accept'][it], bool(rec['accept'][0])
        margin = abs(float(U[it]) - min(1.0, float(np.exp(-trace['dH'][it]))))
        assert a == b or margin < 1e-3, (it, a, b, margin, trace['dH'][it], float(rec['dH'][0]))
        assert int(rec['epoch'][0]) == trace['epoch'][it]
    err = rel(res.samples[0], want)
    print(f'mass loop: returned samples rel err {err:.2e}')
    assert res.samples.shape[1] == 35 and err < 1e-3
