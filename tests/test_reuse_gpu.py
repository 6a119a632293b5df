"""GPU: the gradient cache (one score ladder per leapfrog step: the first half step reuses the previous evaluation,
main_sampling.py:693-695 vs :709-711) and the compaction of finished chains.  Both must change NO bit of a run:
the cached kernels are compared with the plain ones on the same operands, whole runs with the two features on / off."""
import copy
import types

import numpy as np
import pytest
import torch

from oracle import operators as oops, schedule as osched

pytestmark = pytest.mark.gpu
SEQ, SEQ_NEXT = [250, 500, 750], [-1, 250, 500]


def gen(seed):
    return torch.Generator().manual_seed(seed)


class F64Score(torch.nn.Module):
    """tiny score in float64, rounded to fp32: convolution results then do not depend on the batch they ran in"""

    def __init__(self, net):
        super().__init__()
        self.net = copy.deepcopy(net).double()

    def forward(self, x, t):
        return self.net(x.double(), t.double()).float()


@pytest.mark.parametrize('with_g2', [True, False])
def test_cached_first_and_last_steps_are_the_plain_kernels(with_g2):
    import nhmc.kernels as K
    B, shape = 5, (3, 32, 32)
    N = int(np.prod(shape))
    g_ = gen(70)
    dev = 'cuda'
    x = torch.randn(B, *shape, generator=g_).to(dev)
    p = torch.randn(B, *shape, generator=g_).to(dev)
    ga = torch.randn(B, *shape, generator=g_).to(dev) * 30
    gb = (torch.randn(B, *shape, generator=g_).to(dev) * 30) if with_g2 else None
    loss = (torch.rand(B, generator=g_).double() * 1e4).to(dev)
    eps = torch.tensor([0.05, 0.04, 0.0, 0.01, 0.05], dtype=torch.float64, device=dev)
    sig = torch.tensor([1.7, 0.9, 0.1, 0.5, 1.0], dtype=torch.float64, device=dev)
    sel = torch.tensor([0, 1, 1, 0, 1], dtype=torch.int32, device=dev)
    g_pair = torch.full((2, B) + shape, float('nan'), device=dev)
    loss_pair = torch.full((2, B), float('nan'), dtype=torch.float64, device=dev)

    # store at slot sel: the summed gradient and the loss, nothing in the other slot
    K.grad_cache_store(ga, gb, loss, g_pair, loss_pair, sel)
    gsum = ga + gb if with_g2 else ga
    for c in range(B):
        s = int(sel[c])
        assert torch.equal(g_pair[s, c], gsum[c]) and bool(torch.isnan(g_pair[1 - s, c]).all())
        assert float(loss_pair[s, c]) == float(loss[c]) and bool(torch.isnan(loss_pair[1 - s, c]))

    # FIRST from the cache == FIRST with (g, g2)
    ws_a, ws_b = K.leapfrog_ws(B, N, dev), K.leapfrog_ws(B, N, dev)
    xa, pa = torch.empty_like(x), p.clone()
    K.leapfrog_first(x, xa, pa, ga, eps, sig, 1.0, ws_a, g2=gb)
    xb, pb = torch.empty_like(x), p.clone()
    K.leapfrog_first_cached(x, xb, pb, g_pair, sel, eps, sig, 1.0, ws_b)
    assert torch.equal(xa, xb) and torch.equal(pa, pb) and torch.equal(ws_a, ws_b)
    H_a = K.hamiltonian(ws_a, N, loss, sig, 1.0)
    H_b = K.hamiltonian(ws_b, N, loss_pair, sig, 1.0, sel=sel)
    assert torch.equal(H_a, H_b)

    # LAST with the cache == plain LAST, and the OTHER slot receives the end point's (g + g2, loss)
    g2a = torch.randn(B, *shape, generator=g_).to(dev) * 30
    g2b = (torch.randn(B, *shape, generator=g_).to(dev) * 30) if with_g2 else None
    loss2 = (torch.rand(B, generator=g_).double() * 1e4).to(dev)
    pa2, pb2 = pa.clone(), pa.clone()
    K.leapfrog_fused(K.LF_LAST, xa, pa2, g2a, eps, sig, 1.0, ws_a, g2=g2b)
    K.leapfrog_last_cached(xb, pb2, g2a, g2b, g_pair, loss_pair, sel, loss2, eps, sig, 1.0, ws_b)
    assert torch.equal(pa2, pb2) and torch.equal(xa, xb) and torch.equal(ws_a, ws_b)
    gsum2 = g2a + g2b if with_g2 else g2a
    for c in range(B):
        s = int(sel[c])
        assert torch.equal(g_pair[s, c], gsum[c]) and torch.equal(g_pair[1 - s, c], gsum2[c])
        assert float(loss_pair[s, c]) == float(loss[c]) and float(loss_pair[1 - s, c]) == float(loss2[c])

    # chunk views: chains [2, 5) only
    g_pair2 = torch.zeros_like(g_pair)
    loss_pair2 = torch.zeros_like(loss_pair)
    K.grad_cache_store(ga[2:], gb[2:] if with_g2 else None, loss[2:], g_pair2[:, 2:], loss_pair2[:, 2:], sel[2:], flip=1)
    for c in range(B):
        s = 1 - int(sel[c])
        want = gsum[c] if c >= 2 else torch.zeros_like(gsum[c])
        assert torch.equal(g_pair2[s, c], want) and float(g_pair2[1 - s, c].abs().max()) == 0.0
        assert float(loss_pair2[s, c]) == (float(loss[c]) if c >= 2 else 0.0)

    accept = torch.tensor([1, 0, 1, 0, 0], dtype=torch.int32, device=dev)
    sel2 = sel.clone()
    K.grad_cache_flip(accept, sel2)
    assert sel2.tolist() == [1, 1, 0, 0, 1]


def _problem(tiny_score, B, dim=16, seed=71, f64=True):
    import nhmc.operators as ops
    from nhmc import plugin
    g_ = gen(seed)
    missing = oops.random_inpaint_missing(dim, generator=g_)
    op = ops.Inpainting(3, dim, missing, 'cuda')
    ref = oops.InpaintRef(3, dim, missing)
    x = torch.randn(B, 3, dim, dim, generator=g_)
    x_orig = torch.rand(B, 3, dim, dim, generator=g_) * 2 - 1
    y = ref.H(x_orig) + 0.1 * torch.randn(B, ref.M, generator=g_)
    net = (F64Score(tiny_score) if f64 else copy.deepcopy(tiny_score)).cuda()
    algo = plugin.HMC(net, op, 0.1)
    return algo, op, x.cuda(), y.cuda(), x_orig.cuda()


def _same_run(a, b):
    assert a.iters == b.iters
    for k in ('samples', 'x', 'xt', 'epoch', 'n_accept', 'n_reject', 'psnr'):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    for ra, rb in zip(a.trace, b.trace):
        for k in ('accept', 'epoch', 'sigma_y', 'eps'):
            assert torch.equal(ra[k], rb[k]), k


@pytest.mark.parametrize('chunk', [None, 2])
def test_whole_run_with_the_gradient_cache_is_bit_identical_and_saves_one_ladder_per_trajectory(tiny_score, chunk):
    """accepts, rejects, the tau / eps anneal after repeated rejects and the sigma_y schedule all occur in this run"""
    from nhmc import sampler
    B = 3
    algo, op, x, y, x_orig = _problem(tiny_score, B)
    b = osched.betas_fp32().cuda()
    opt = types.SimpleNamespace(tau=0.3, epsilon=0.05, m=1.0, sigma_0=0.1)
    kw = dict(epochs=6, sampling=2, collect_trace=True, chunk=chunk, compact=False)
    runs = {}
    for reuse in (False, True):
        runs[reuse] = sampler.hmc_chains(x, b, SEQ, SEQ_NEXT, algo, opt, y, op, x_orig, noise=sampler.PhiloxNoise(5, 0),
                                         reuse=reuse, **kw)
    old, new = runs[False], runs[True]
    _same_run(old, new)
    for ra, rb in zip(old.trace, new.trace):
        assert torch.equal(ra['dH'], rb['dH'])
    n_rej = int(new.n_reject.sum())
    assert n_rej > 0 and int(new.n_accept.min()) == 10                      # both branches of the cache were taken
    L, chunks = new.L, (1 if chunk is None else -(-B // chunk))
    assert L == 5                                                           # floor(0.3 / 0.05) in floating point
    assert old.ladders == old.iters * (L + 1) * chunks
    assert new.ladders == (new.iters * L + 1) * chunks                     # L per trajectory + the start point, once
    assert new.chain_ladders == (new.iters * L + 1) * B


def test_finished_chains_leave_the_batch_without_changing_any_chain(tiny_score):
    """six chains that finish at different trajectories: with compaction every chain's samples / position / counters
    are the uncompacted run's bits (the score runs in float64, so its output does not depend on the batch it is
    evaluated in), and fewer chains go through the score network."""
    from nhmc import sampler
    B = 6
    algo, op, x, y, x_orig = _problem(tiny_score, B, seed=72)
    b = osched.betas_fp32().cuda()
    opt = types.SimpleNamespace(tau=0.2, epsilon=0.05, m=1.0, sigma_0=0.1)
    kw = dict(epochs=5, sampling=2, collect_trace=True)
    full = sampler.hmc_chains(x, b, SEQ, SEQ_NEXT, algo, opt, y, op, x_orig, noise=sampler.PhiloxNoise(9, 0),
                              compact=False, **kw)
    for quantum, chunk in ((1, None), (2, None), (None, 2)):
        comp = sampler.hmc_chains(x, b, SEQ, SEQ_NEXT, algo, opt, y, op, x_orig, noise=sampler.PhiloxNoise(9, 0),
                                  compact=True, compact_quantum=quantum, chunk=chunk, **kw)
        _same_run(full, comp)
        finish = [max(i for i, r in enumerate(full.trace) if int(r['epoch'][c]) < 9) for c in range(B)]
        assert len(set(finish)) > 2, finish                                 # staggered finishes, or the test shows nothing
        for i, (ra, rb) in enumerate(zip(full.trace, comp.trace)):         # energies of every chain while it runs
            for c in range(B):
                if i <= finish[c]:
                    assert float(ra['dH'][c]) == float(rb['dH'][c]), (i, c)
        assert comp.chain_trajectories < full.chain_trajectories == full.iters * B
        q = quantum or chunk
        want = sum(min(B, -(-sum(1 for c in range(B) if finish[c] >= i) // q) * q) for i in range(full.iters))
        assert comp.chain_trajectories == want, (comp.chain_trajectories, want)
        assert comp.chain_ladders == comp.chain_trajectories * comp.L + B


def test_latent_loop_with_the_gradient_cache_is_bit_identical(tiny_score):
    """hmc_latent_chains (score without gradient, image map between decode and operator) with / without the cache"""
    from nhmc import sampler, operators as ops, plugin

    class TinyLatent(torch.nn.Module):
        def __init__(self, net):
            super().__init__()
            self.net = F64Score(net)
            ab = torch.cumprod(1 - torch.linspace(1e-4, 0.02, 1000, dtype=torch.float64), 0).float()
            self.register_buffer('alphas_cumprod', ab)
            self.register_buffer('alphas_cumprod_prev', torch.cat([torch.ones(1), ab[:-1]]))
            self.up = torch.nn.Upsample(scale_factor=2, mode='nearest')

        def apply_model(self, x, t, c=None):
            with torch.no_grad():
                return self.net(x, t)[:, :3]

        def differentiable_decode_first_stage(self, z):
            return torch.tanh(self.up(z) * 0.9)

    dim, B = 8, 3
    g_ = gen(73)
    model = TinyLatent(tiny_score).cuda()
    missing = oops.random_inpaint_missing(2 * dim, generator=g_)
    op = ops.Inpainting(3, 2 * dim, missing, 'cuda')
    algo = plugin.HMCLatent(model, op, 0.1)
    x = torch.randn(B, 3, dim, dim, generator=g_).cuda()
    y = (torch.randn(B, op.M, generator=g_) * 0.5).cuda()
    opt = types.SimpleNamespace(tau=0.3, epsilon=0.1, m=1.0, sigma_0=0.1, sigma_y=0.5)
    runs = [sampler.hmc_latent_chains(x, SEQ, SEQ_NEXT, algo, opt, y, op, noise=sampler.PhiloxNoise(3, 0), epochs=5, sampling=2,
                                      collect_trace=True, reuse=r) for r in (False, True)]
    assert torch.equal(runs[0].x, runs[1].x) and torch.equal(runs[0].xt, runs[1].xt)
    for ra, rb in zip(runs[0].trace, runs[1].trace):
        assert torch.equal(ra['dH'], rb['dH']) and torch.equal(ra['accept'], rb['accept'])
    for sa, sb in zip(runs[0].samples, runs[1].samples):
        assert torch.equal(sa, sb)
    L = runs[0].L
    assert runs[0].ladders == 9 * (L + 1) and runs[1].ladders == 9 * L + 1
