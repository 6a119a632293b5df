"""GPU: the sampler (decode + hand-unrolled backward, trajectories, the per-chain outer loop)
against the CPU oracle and the reference-captured goldens.  Tolerance: north_star's 1e-4 relative
for images / positions; Hamiltonians to a few fp32 ulps of their ~1e3..1e5 magnitude."""
import copy
import types

import numpy as np
import pytest
import torch

from oracle import hmc_ref, operators as oops, schedule as osched

pytestmark = pytest.mark.gpu
SEQ, SEQ_NEXT = [250, 500, 750], [-1, 250, 500]
T = torch.from_numpy


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def gen(seed):
    return torch.Generator().manual_seed(seed)


def make_ops(deg, dim, missing=None, aniso=None):
    import nhmc.operators as ops
    if deg == 'inpaint':
        return oops.InpaintRef(3, dim, missing), ops.Inpainting(3, dim, missing, 'cuda')
    if deg.startswith('sr'):
        r = int(deg[2:])
        return oops.BlockMeanRef(3, dim, r), ops.SuperResolution(3, dim, r, 'cuda')
    ref = oops.SpectralBlurRef(*aniso) if aniso is not None else \
        oops.SpectralBlurRef.from_kernels(oops.gaussian_taps(1.0), oops.gaussian_taps(20.0), 3, dim)
    return ref, ops.Deblurring2D.from_factors(ref.U1, ref.U2, ref.V1, ref.V2, ref.D, 'cuda')


def engine_for(net_gpu, op, chunk=None):
    from nhmc import plugin, sampler
    algo = plugin.HMC(net_gpu, op, 0.1)
    return algo, sampler.LeapfrogEngine(algo.score, op, osched.betas_fp32().cuda(), SEQ, SEQ_NEXT, torch.device('cuda'),
                                        chunk=chunk)


def state_for(B, eps, sigma_y):
    from nhmc import sampler
    st = sampler.ChainState(B, 1.0, 0.05, 'cuda')
    st['eps_eff'].copy_(torch.as_tensor(np.broadcast_to(np.asarray(eps, dtype=np.float64), (B,)).copy()))
    st['sigma_y'].copy_(torch.as_tensor(np.broadcast_to(np.asarray(sigma_y, dtype=np.float64), (B,)).copy()))
    return st


@pytest.mark.parametrize('deg', ['inpaint', 'sr4', 'aniso'])
@pytest.mark.parametrize('chunk', [None, 2])
def test_decode_and_gradient_match_oracle(tiny_score, deg, chunk):
    dim, B = 32, 3
    g_ = gen(20)
    missing = oops.random_inpaint_missing(dim, generator=g_)
    ref, op = make_ops(deg, dim, missing)
    x = torch.randn(B, 3, dim, dim, generator=g_)
    y = torch.randn(B, ref.M, generator=g_)
    leaf = x.clone().requires_grad_(True)
    xt_r, _, loss_r, grad_r = hmc_ref._data_loss_and_grad(leaf, osched.betas_fp32(), SEQ, SEQ_NEXT, tiny_score, ref, y)
    _, eng = engine_for(copy.deepcopy(tiny_score).cuda(), op, chunk)
    xt, loss, ga, gb = eng.decode_and_grad(x.cuda(), y.cuda())
    assert rel(xt, xt_r) < 1e-5 and rel(loss, loss_r) < 1e-5
    assert rel(ga + gb, grad_r) < 1e-4
    assert rel(eng.decode(x.cuda()), xt_r) < 1e-5


def test_plugin_surface_gives_the_same_decode_and_gradient(tiny_score):
    """cal_x0 / map_back through autograd (the reference's iterative_sampling) == the fused engine."""
    from nhmc import sampler
    dim, B = 32, 2
    g_ = gen(21)
    missing = oops.random_inpaint_missing(dim, generator=g_)
    ref, op = make_ops('inpaint', dim, missing)
    algo, eng = engine_for(copy.deepcopy(tiny_score).cuda(), op)
    x = torch.randn(B, 3, dim, dim, generator=g_).cuda()
    y = torch.randn(B, ref.M, generator=g_).cuda()
    leaf = x.clone().requires_grad_(True)
    opt = types.SimpleNamespace(algo='hmc', noise='ddpm')
    xt = sampler.iterative_sampling(leaf, B, osched.betas_fp32().cuda(), SEQ, SEQ_NEXT, algo, opt, y).clip(-1, 1)
    # the operator's H() is a raw kernel (not an autograd op); the fused data term supplies d loss / d xt
    l, g_xt = op.data_term(xt.detach().contiguous(), y, apply_clip=True)
    (grad,) = torch.autograd.grad(xt, leaf, g_xt)
    xt2, l2, ga, gb = eng.decode_and_grad(x, y)
    assert torch.equal(xt.detach(), xt2) and torch.equal(l, l2)
    assert rel(grad, ga + gb) < 1e-6
    assert algo.et.shape == (B, 3, dim, dim)


@pytest.mark.parametrize('deg', ['inpaint', 'sr4', 'aniso'])
def test_trajectory_matches_oracle(tiny_score, deg):
    from nhmc import sampler
    dim, B, L = 32, 3, 20
    g_ = gen(22)
    missing = oops.random_inpaint_missing(dim, generator=g_)
    ref, op = make_ops(deg, dim, missing)
    x, p = torch.randn(B, 3, dim, dim, generator=g_), torch.randn(B, 3, dim, dim, generator=g_)
    y = ref.H(torch.rand(B, 3, dim, dim, generator=g_) * 2 - 1) + 0.1 * torch.randn(B, ref.M, generator=g_)
    eps, sig = np.array([0.05, 0.05 * 0.95, 0.01]), np.array([1.7, 0.9, 0.1])
    want = hmc_ref.trajectory(x, p.clone(), osched.betas_fp32(), SEQ, SEQ_NEXT, tiny_score, ref, y,
                              sigma_y=sig, eps=eps, m=1.0, L=L)
    _, eng = engine_for(copy.deepcopy(tiny_score).cuda(), op)
    got = sampler.run_trajectory(eng, x.cuda(), p.cuda().clone(), y.cuda(), state_for(B, eps, sig), 1.0, L)
    assert rel(got['x_prop'], want['x']) < 1e-4 and rel(got['p'], want['p']) < 1e-4
    assert rel(got['xt'], want['xt']) < 1e-4 and rel(got['loss'], want['loss']) < 1e-4
    for k in ('H0', 'H1'):
        ulp = float(np.spacing(np.float32(want[k].abs().max())))
        assert float((got[k].cpu() - want[k]).abs().max()) <= max(8 * ulp, 1e-4 * float(want[k].abs().max()) * 0.01)


@pytest.mark.parametrize('deg', ['inpaint', 'sr4', 'aniso'])
def test_first_trajectory_of_the_reference_run(golden, tiny_score, deg):
    """Reference-captured first outer iteration (G4): same x0, p0, y -> same end position, decode, dH."""
    from nhmc import sampler
    g = golden(f'g4_hmc_{deg}_32.npz')
    aniso = [T(g[f'aniso_{k}']) for k in ('U1', 'U2', 'V1', 'V2', 'D')] if deg == 'aniso' else None
    _, op = make_ops(deg, 32, T(g['missing']), aniso)
    _, eng = engine_for(copy.deepcopy(tiny_score).cuda(), op)
    sig = hmc_ref.sigma_y_at(0, float(g['sigma_0']))
    got = sampler.run_trajectory(eng, T(g['x']).cuda(), T(g['p0']).cuda().clone(), T(g['y_0']).cuda(),
                                 state_for(1, 0.05, sig), 1.0, 20)
    assert rel(got['x_prop'], T(g['pos_last'])) < 1e-4
    assert rel(got['xt'], T(g['dec_last'])) < 1e-4
    dH = float((got['H1'] - got['H0'])[0])
    assert abs(dH - (-float(g['neg_dH'][0]))) < 0.05


def test_outer_loop_matches_oracle_chains(tiny_score):
    """Whole per-chain loop (schedules, accept commit, sample collection) vs oracle.hmc_chains on the
    same noise tape: same accept decisions (or an ambiguous threshold), same collected samples."""
    from nhmc import sampler
    dim, B = 16, 3
    g_ = gen(23)
    missing = oops.random_inpaint_missing(dim, generator=g_)
    ref, op = make_ops('inpaint', dim, missing)
    x = torch.randn(B, 3, dim, dim, generator=g_)
    x_orig = torch.rand(B, 3, dim, dim, generator=g_) * 2 - 1
    y = ref.H(x_orig) + 0.1 * torch.randn(B, ref.M, generator=g_)
    P = [torch.randn(B, 3, dim, dim, generator=g_) for _ in range(64)]
    U = [torch.rand(B, generator=g_) for _ in range(64)]
    kw = dict(epochs=3, sampling=2)
    tr = {}

    class F64Score(torch.nn.Module):
        """The tiny score evaluated in fp64 and rounded to fp32: CPU (oneDNN) and GPU convolutions then agree
        to ~1e-15 before the rounding, so the many-trajectory comparison below is sensitive to the sampler's
        kernels and not to the chaotic amplification of conv-implementation noise."""

        def __init__(self, net):
            super().__init__()
            self.net = copy.deepcopy(net).double()

        def forward(self, x, t):
            return self.net(x.double(), t.double()).float()

    cpu_score = F64Score(tiny_score)
    want = hmc_ref.hmc_chains(x, osched.betas_fp32(), SEQ, SEQ_NEXT, cpu_score, ref, y, x_orig, tau=1.0, epsilon=0.05,
                              m=1.0, sigma_0=0.1, draw_p=lambda it: P[it], draw_u=lambda it: U[it], trace=tr, **kw)
    algo, _ = engine_for(F64Score(tiny_score).cuda(), op)
    opt = types.SimpleNamespace(tau=1.0, epsilon=0.05, m=1.0, sigma_0=0.1)
    res = sampler.hmc_chains(x.cuda(), osched.betas_fp32().cuda(), SEQ, SEQ_NEXT, algo, opt, y.cuda(), op, x_orig.cuda(),
                             noise=sampler.TapeNoise(lambda it: P[it], lambda it: U[it]), collect_trace=True, **kw)
    assert res.iters == len(tr['accept'])
    for it, rec in enumerate(res.trace):
        want_acc, got_acc = tr['accept'][it], rec['accept'].numpy().astype(bool)
        margin = np.abs(U[it].numpy() - np.minimum(1.0, np.exp(-tr['dH'][it])))
        assert np.all((want_acc == got_acc) | (margin < 1e-3)), (it, want_acc, got_acc, margin)
        assert np.array_equal(rec['epoch'].numpy(), tr['epoch'][it])
    assert res.epoch.cpu().tolist() == [7] * B
    assert rel(res.samples, want) < 1e-4


def test_reference_entry_point_shapes_and_shard_invariance():
    """hmc(...) keeps the reference's return shape at n = 1; with Philox noise a chain's result does not
    depend on which launch (shard) it ran in."""
    from nhmc import operators, plugin, sampler

    class PointwiseScore(torch.nn.Module):           # no cross-sample / cross-pixel op: bitwise batch-invariant
        def forward(self, x, t):
            a = (t / 1000.0).view(-1, 1, 1, 1)
            e = torch.tanh(x * 0.7) * (0.5 + a)
            return torch.cat([e, torch.zeros_like(e)], dim=1)

    dim = 16
    dev = torch.device('cuda')
    op = operators.build_operator('inpaint_random', 3, dim, dev, generator=gen(24))
    algo = plugin.HMC(PointwiseScore().to(dev), op, 0.1)
    b = osched.betas_fp32().to(dev)
    g_ = gen(25)
    x = torch.randn(4, 3, dim, dim, generator=g_).to(dev)
    x_orig = (torch.rand(4, 3, dim, dim, generator=g_) * 2 - 1).to(dev)
    y = op.H(x_orig) + 0.1 * torch.randn(4, op.M, generator=g_).to(dev)
    opt = types.SimpleNamespace(tau=0.2, epsilon=0.05, m=1.0, sigma_0=0.1, quiet=True)
    kw = dict(epochs=2, sampling=1, max_iters=12)
    full = sampler.hmc_chains(x, b, SEQ, SEQ_NEXT, algo, opt, y, op, x_orig, noise=sampler.PhiloxNoise(5678, 0), **kw)
    lo = sampler.hmc_chains(x[:2], b, SEQ, SEQ_NEXT, algo, opt, y[:2], op, x_orig[:2], noise=sampler.PhiloxNoise(5678, 0), **kw)
    hi = sampler.hmc_chains(x[2:], b, SEQ, SEQ_NEXT, algo, opt, y[2:], op, x_orig[2:], noise=sampler.PhiloxNoise(5678, 2), **kw)
    # shards may stop at different iteration counts; compare what both ran: final state of finished chains
    done = (full.epoch.cpu() >= 4)
    parts_x = torch.cat([lo.x, hi.x])
    parts_s = torch.cat([lo.samples, hi.samples])
    for c in range(4):
        if bool(done[c]) and int(torch.cat([lo.epoch, hi.epoch])[c]) >= 4:
            assert torch.equal(full.samples[c], parts_s[c]) and torch.equal(full.x[c], parts_x[c])
    opt1 = types.SimpleNamespace(tau=0.2, epsilon=0.05, m=1.0, sigma_0=0.1, quiet=True, philox_seed=1)
    out = sampler.hmc(x[:1], 1, b, SEQ, SEQ_NEXT, algo, opt1, y[:1], op, x_orig[:1])
    assert out.shape == (20, 3, dim, dim) and bool(torch.isfinite(out).all())
    assert float(out.abs().max()) <= 1.0


def test_cli_runs_the_reference_command_line(tmp_path, monkeypatch, capsys):
    """`main_sampling.py`-compatible flags end to end on a small config (32x32 U-Net of the same architecture)."""
    import yaml
    from nhmc import cli
    cfgdir = tmp_path / 'configs'
    cfgdir.mkdir()
    cfg = {'data': {'dataset': 'tiny', 'image_size': 32, 'channels': 3, 'rescaled': True},
           'model': dict(image_size=32, num_channels=32, num_res_blocks=1, channel_mult='1,2', learn_sigma=True,
                         class_cond=False, use_checkpoint=False, attention_resolutions='16', num_heads=4,
                         num_head_channels=16, num_heads_upsample=-1, use_scale_shift_norm=True, dropout=0.0,
                         resblock_updown=True, use_fp16=False, use_new_attention_order=False, model_path=''),
           'diffusion': {'beta_schedule': 'linear', 'beta_start': 1e-4, 'beta_end': 0.02, 'num_diffusion_timesteps': 1000}}
    (cfgdir / 'config_tiny.yml').write_text(yaml.safe_dump(cfg))
    monkeypatch.chdir(tmp_path)
    table = cli.main(['--dataset', 'tiny', '--algo', 'hmc', '--timesteps', '3', '--deg', 'sr4', '--sigma_0', '0.05',
                      '-i', str(tmp_path / 'out'), '--tau', '0.1', '--epsilon', '0.05', '--synthetic', '2', '--chains', '2',
                      '--philox', '--ni', '--doc', 'ignored'])
    assert table.shape == (2, 3) and bool(torch.isfinite(table).all())
    assert 'Total Average PSNR' in capsys.readouterr().out
    with pytest.raises(NotImplementedError):
        cli.main(['--dataset', 'tiny', '--algo', 'dps', '--deg', 'sr4', '--sigma_0', '0.05'])


@pytest.mark.parametrize('deg', ['inpaint', 'sr4', 'aniso'])
def test_trajectory_does_not_depend_on_the_score_chunking(tiny_score, deg):
    """`LeapfrogEngine.step` hands each score chunk's gradient pieces straight to that chunk's fused update (FIRST out of
    place, then in place): chunked and unchunked trajectories are the same bits for a score without cross-sample ops,
    and the caller's position is left untouched."""
    from nhmc import sampler
    dim, B, L = 32, 5, 4
    g_ = gen(41)
    missing = oops.random_inpaint_missing(dim, generator=g_)
    ref, op = make_ops(deg, dim, missing)
    x = torch.randn(B, 3, dim, dim, generator=g_).cuda()
    p = torch.randn(B, 3, dim, dim, generator=g_).cuda()
    y = (ref.H(torch.rand(B, 3, dim, dim, generator=g_) * 2 - 1) + 0.1 * torch.randn(B, ref.M, generator=g_)).cuda()
    eps, sig = np.array([0.05, 0.04, 0.03, 0.0, 0.05]), np.array([1.7, 0.9, 0.1, 0.5, 1.0])     # chain 3 frozen (eps_eff = 0)
    outs = []
    for chunk in (None, 2, 3):
        _, eng = engine_for(copy.deepcopy(tiny_score).cuda(), op, chunk=chunk)
        x0 = x.clone()
        got = sampler.run_trajectory(eng, x0, p.clone(), y, state_for(B, eps, sig), 1.0, L)
        assert torch.equal(x0, x)                                              # accepted position untouched
        outs.append({k: got[k].clone() for k in ('x_prop', 'p', 'xt', 'loss', 'H0', 'H1')})
    for other in outs[1:]:
        for k, v in outs[0].items():
            assert torch.equal(v, other[k]) or rel(other[k], v) < 1e-6, k
    assert torch.equal(outs[0]['x_prop'][3], x[3])                             # the frozen chain did not move
