"""GPU: the spectral data term with the residual taken in the operator's left singular basis (four products,
nhmc_data_spectral_proj) against the reference's eight-product sequence (nhmc_data_spectral) and against the reference
formula evaluated in float64 (obs_functions/Hfuncs.py:448-523 + main_sampling.py:694-695,710-711)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def _f64_data_term(op, x, y):
    """loss = |y - H clip(x)|^2 and its x-gradient with the operator's own fp32 factors, every product in float64."""
    U1, U2, V1, V2 = (op.factors[i].double() for i in range(4))
    D = op.Dmap.double()
    xc = x.double().clip(-1, 1)
    r = y.double().reshape(x.shape) - U1 @ (D * (V1.t() @ xc @ V2)) @ U2.t()
    g = -2 * (V1 @ (D * (U1.t() @ r @ U2)) @ V2.t()) * (x.abs() <= 1)
    return (r * r).sum((1, 2, 3)), g


@pytest.mark.parametrize('dim,B', [(256, 4), (64, 3), (32, 2)])
@pytest.mark.parametrize('sigma0', [0.05, 0.01])
def test_projected_form_against_the_eight_products_and_float64(dim, B, sigma0):
    from nhmc import kernels as K, operators
    op = operators.build_operator('deblur_aniso', 3, dim, 'cuda', spectral_projected=True)
    assert op.projected and op.orthogonality_error < 1e-5
    g = torch.Generator().manual_seed(dim + int(1000 * sigma0))
    x0 = torch.nn.functional.avg_pool2d(torch.rand(B, 3, dim + 4, dim + 4, generator=g) * 2 - 1, 5, 1)
    y = op.H(x0.cuda()) + sigma0 * torch.randn(B, 3 * dim * dim, generator=g).cuda()
    # a state next to the truth: |r| is at the noise level, where the missing U^T U cancellation weighs most; some
    # entries beyond [-1, 1] so the clip mask is exercised
    x = (1.02 * x0 + 0.05 * torch.randn(x0.shape, generator=g)).cuda().contiguous()
    yT = y.reshape(x.shape).transpose(-1, -2).contiguous()                # the eight-product kernels take the planes transposed
    loss8, g8 = K.data_spectral(x, yT, op.factors, op.Dmap, True, DmapT=op.DmapT)
    loss4, g4 = op.data_term(x, y, True)
    loss64, g64 = _f64_data_term(op, x, y)
    e8, e4 = _rel(g8, g64), _rel(g4, g64)
    print(f'd={dim} sigma0={sigma0}: gradient vs float64: eight products {e8:.2e}, four {e4:.2e}; '
          f'loss {_rel(loss8, loss64):.2e} / {_rel(loss4, loss64):.2e}')
    assert e8 < 2e-5 and e4 < 3e-5
    assert _rel(loss8, loss64) < 1e-5 and _rel(loss4, loss64) < 1e-5
    assert float((g4 - g8).abs().max() / g8.abs().max()) < 5e-5
    assert ((g4 == 0) == (g8 == 0))[x.abs() > 1].all()


def test_projected_vjp_form_and_the_observation_cache():
    from nhmc import kernels as K, operators
    op = operators.build_operator('deblur_aniso', 3, 256, 'cuda', spectral_projected=True)
    B = 3
    x = K.randn_philox((B, 3, 256, 256), 5, 0, 0)
    e = K.randn_philox((B, 6, 256, 256), 5, 0, 1)
    y = op.H(0.3 * K.randn_philox((B, 3, 256, 256), 5, 0, 2))
    at, an = torch.full((B,), 0.52, device='cuda'), torch.ones(B, device='cuda')
    nxt = K.ddim_mix_fwd(x, e, at, an, final_clip=True)['xt_next']
    yT = y.reshape(x.shape).transpose(-1, -2).contiguous()
    l8, gx8, ge8 = K.data_spectral_vjp(nxt, yT, op.factors, op.Dmap, x, e, at, an, DmapT=op.DmapT)
    l4, gx4, ge4 = op.fused_last_vjp(x, e, at, an, y, xt_next=nxt)
    assert _rel(l4, l8) < 1e-5 and _rel(gx4, gx8) < 3e-5 and _rel(ge4[:, :3], ge8[:, :3]) < 3e-5
    assert (ge4[:, 3:] == 0).all()
    # same buffer, same version -> one projection; an in-place write -> a new one
    n0 = len(op._y_proj)
    op.fused_last_vjp(x, e, at, an, y, xt_next=nxt)
    assert len(op._y_proj) == n0
    y.mul_(0.5)
    l4b, _, _ = op.fused_last_vjp(x, e, at, an, y, xt_next=nxt)
    l8b, _, _ = K.data_spectral_vjp(nxt, y.reshape(x.shape).transpose(-1, -2).contiguous(), op.factors, op.Dmap, x, e, at, an,
                                    DmapT=op.DmapT)
    assert len(op._y_proj) == n0 + 1 and _rel(l4b, l8b) < 1e-5


def test_projected_pair_launches_equal_the_one_product_chain(monkeypatch):
    """d = 256: two products per launch (k_pair256) and the one-product-per-launch chain, which parks its second
    intermediate in g_xt, run the same MFMA sequence: same bits."""
    from nhmc import kernels as K, operators
    op = operators.build_operator('deblur_aniso', 3, 256, 'cuda', spectral_projected=True)
    B = 2
    x = 0.7 * K.randn_philox((B, 3, 256, 256), 9, 0, 0)
    e = K.randn_philox((B, 6, 256, 256), 9, 0, 1)
    y = op.H(0.3 * K.randn_philox((B, 3, 256, 256), 9, 0, 2))
    at, an = torch.full((B,), 0.52, device='cuda'), torch.ones(B, device='cuda')
    nxt = K.ddim_mix_fwd(x, e, at, an, final_clip=True)['xt_next']
    out = {}
    for pairs in ('1', '0'):
        monkeypatch.setenv('NHMC_SPECTRAL_PAIRS', pairs)
        op._y_proj.clear()
        out[pairs] = op.data_term(x, y, True) + op.fused_last_vjp(x, e, at, an, y, xt_next=nxt)
    torch.cuda.synchronize()
    for a, b in zip(out['1'], out['0']):
        assert torch.equal(a, b) or _rel(a, b) < 1e-6


def test_the_default_is_the_reference_sequence(monkeypatch):
    from nhmc import operators
    monkeypatch.delenv('NHMC_SPECTRAL_PROJECTED', raising=False)
    assert not operators.build_operator('deblur_aniso', 3, 64, 'cuda').projected
    monkeypatch.setenv('NHMC_SPECTRAL_PROJECTED', '1')
    assert operators.build_operator('deblur_aniso', 3, 64, 'cuda').projected


def test_projected_form_on_the_reference_run(golden, tiny_score):
    """The G14 replay (tests/test_reference_run_gpu.py) with the four-product form.  The reference's loss is not smooth
    (the final clip masks the gradient), so an implementation that rounds differently follows the reference's run only
    until a decode lands within its deviation of +-1: the CPU restatement of this form departs at trajectory 130 (one
    pixel's mask bit), the eight-product form stays on the reference for all 191.  Checked: the first 120 trajectories
    (2500 leapfrog steps) reproduce the reference's accept decisions and its energy differences."""
    import types
    import numpy as np
    from nhmc import operators, plugin, sampler
    from oracle import schedule as osched
    from oracle.tiny_score import F64Score
    from tests.test_reference_run_gpu import BAND, SEQ, SEQ_NEXT, T, tape_of
    g = golden('g14_hmc_f64_aniso_32.npz')
    dev = torch.device('cuda')
    op = operators.Deblurring2D.from_factors(*(T(g[f'aniso_{k}']) for k in ('U1', 'U2', 'V1', 'V2', 'D')), dev, projected=True)
    P, N = tape_of(g), 120
    prob = np.minimum(1.0, np.exp(np.minimum(g['neg_dH'], 50.0)))
    ref_acc = g['u'] < prob
    u_play = np.where(np.abs(g['u'] - prob) < BAND, np.where(ref_acc, 0.0, 1.0), g['u']).astype(np.float32)
    algo = plugin.HMC(F64Score(tiny_score).to(dev), op, float(g['sigma_0']))
    opt = types.SimpleNamespace(tau=float(g['tau']), epsilon=float(g['epsilon']), m=float(g['m']), sigma_0=float(g['sigma_0']), quiet=True)
    noise = sampler.TapeNoise(lambda it: P[min(it, len(P) - 1)], lambda it: torch.tensor([u_play[min(it, len(P) - 1)]]))
    res = sampler.hmc_chains(T(g['x']).to(dev), osched.betas_fp32().to(dev), SEQ, SEQ_NEXT, algo, opt, T(g['y_0']).to(dev), op,
                             T(g['x_orig']).to(dev), noise=noise, collect_trace=True, max_iters=N)
    got_acc = np.array([bool(r['accept'][0]) for r in res.trace[:N]])
    got_dH = np.array([float(r['dH'][0]) for r in res.trace[:N]])
    assert np.array_equal(got_acc, ref_acc[:N])
    small = np.abs(g['neg_dH'][:N]) < 50
    worst = np.max(np.abs(got_dH[small] + g['neg_dH'][:N][small]))
    print(f'projected form, first {N} trajectories of the reference run: max |dH - dH_ref| = {worst:.4f}')
    assert worst < 0.05


def test_projected_form_under_graph_replay_uses_each_chunks_own_observation(tiny_score):
    """with --graph and the projected form together, the captured graph must contain the projection launch.
    The engine keys its graphs by shape and refills the static observation buffer per chunk; had the capture hit the
    projection cache (filled by the warm-up calls), chunks after the first would be evaluated against the first
    chunk's y^.  Two chunks with different observations, graph on vs off."""
    from nhmc import operators, plugin, sampler
    from oracle import schedule as osched

    class PointwiseScore(torch.nn.Module):                             # capturable: no host tensor is built in forward
        def forward(self, x, t):
            a = (t / 1000.0).view(-1, 1, 1, 1)
            e = torch.tanh(x * 0.7) * (0.5 + a)
            return torch.cat([e, torch.zeros_like(e)], dim=1)

    dev = torch.device('cuda')
    dim, B = 32, 4
    g_ = torch.Generator().manual_seed(77)
    op = operators.build_operator('deblur_aniso', 3, dim, dev, spectral_projected=True)
    assert op.projected
    algo = plugin.HMC(PointwiseScore().to(dev), op, 0.1)
    x = torch.randn(B, 3, dim, dim, generator=g_).to(dev)
    y = (torch.randn(B, op.M, generator=g_)).to(dev)                  # every chain its own observation
    outs = []
    for graph in (False, True, True):                                  # the second graphed call is a pure replay
        eng = sampler.LeapfrogEngine(algo.score, op, osched.betas_fp32().to(dev), [250, 500, 750], [-1, 250, 500], dev, chunk=2) \
            if graph is False or len(outs) == 1 else eng
        xt, loss, ga, gb = eng.decode_and_grad(x, y, graph=graph)
        outs.append((xt.clone(), loss.clone(), ga.clone(), gb.clone()))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            err = float((a.double() - b.double()).abs().max() / a.double().abs().max())
            assert err < 1e-5, err                                     # same kernels; a wrong observation moves the loss by O(1)
    assert float((outs[0][1][:2] - outs[0][1][2:]).abs().min()) > 1e-3 * float(outs[0][1].abs().max())   # the chunks do differ
