"""CPU: pin the oracle (oracle/) against outputs captured from the reference
(tests/golden/*.npz, produced by oracle/gen_golden.py in the build container)."""
import numpy as np
import pytest
import torch

from oracle import schedule, operators, ddim, hmc_ref

T = torch.from_numpy
SEQ, SEQ_NEXT = [250, 500, 750], [-1, 250, 500]


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(1e-30, np.abs(b).max())


def test_g1_schedule_bit_exact(golden):
    g = golden('g1_schedule.npz')
    assert np.array_equal(schedule.beta_schedule(), g['betas64'])
    b = schedule.betas_fp32()
    assert np.array_equal(schedule.alpha_bar_table(b).numpy(), g['table'])
    four = schedule.alpha_bar(b, torch.tensor([750, 500, 250, -1])).reshape(-1).numpy()
    assert np.array_equal(four, g['at_750_500_250_m1'])
    assert four[3] == 1.0
    assert schedule.timestep_ladder(1000, 3) == (SEQ, SEQ_NEXT)


def _ops(g, dim):
    k1, k2 = operators.gaussian_taps(1.0), operators.gaussian_taps(20.0)
    ops = dict(inpaint=operators.InpaintRef(3, dim, T(g['missing'])),
               sr4=operators.BlockMeanRef(3, dim, 4),
               aniso=operators.SpectralBlurRef(*(T(g[f'aniso_{k}']) for k in ('U1', 'U2', 'V1', 'V2', 'D'))))
    if dim % 16 == 0:
        ops['sr16'] = operators.BlockMeanRef(3, dim, 16)
    return ops


@pytest.mark.parametrize('dim', [32, 64])
def test_g2_operators(golden, dim):
    g = golden(f'g2_ops_{dim}.npz')
    x = T(g['x'])
    for name, op in _ops(g, dim).items():
        y = T(g[f'{name}_y'])
        tol = 0.0 if name == 'inpaint' else 2e-6
        assert rel(op.H(x).numpy(), g[f'{name}_Hx']) <= tol, name
        assert rel(op.Ht(y).numpy(), g[f'{name}_Hty']) <= tol, name
        assert rel(op.H_pinv(y).numpy(), g[f'{name}_Hpinvy']) <= max(tol, 5e-6 if name == 'aniso' else tol), name


@pytest.mark.parametrize('dim', [32, 64])
def test_g2_aniso_operator_rebuilt_from_kernels(golden, dim):
    """The multiplier-map construction reproduces the reference's exported D from its own perm."""
    g = golden(f'g2_ops_{dim}.npz')
    D = operators.SpectralBlurRef.multiplier_map(T(g['aniso_s_sorted']), T(g['aniso_perm']), 3, dim)
    assert np.array_equal(D.numpy(), g['aniso_D'])
    # band matrices: the 8-of-9-tap quirk
    Hs = operators.band_matrix(operators.gaussian_taps(20.0), dim)
    assert int((Hs[dim // 2] != 0).sum()) == 8


def test_g3_ddim_decode_and_grad_bit_exact(golden, tiny_score):
    g = golden('g3_ddim_32.npz')
    b = schedule.betas_fp32()
    x = T(g['x']).requires_grad_(True)
    xt = ddim.decode(x, b, SEQ, SEQ_NEXT, tiny_score)
    assert np.array_equal(xt.detach().numpy(), g['xt'])
    grad = torch.autograd.grad((xt.clip(-1, 1) * T(g['w'])).sum(), x)[0]
    assert np.array_equal(grad.numpy(), g['grad'])


def test_g3_ddim_manual_vjp_matches_autograd(golden, tiny_score):
    """Closed-form VJP (what the HIP backward kernel implements) == autograd, bit for bit."""
    g = golden('g3_ddim_32.npz')
    b = schedule.betas_fp32()
    xt = T(g['x']).clone().requires_grad_(True)
    t = torch.ones(2) * 750
    at, at_next = schedule.alpha_bar(b, t.long()), schedule.alpha_bar(b, (torch.ones(2) * 500).long())
    et = tiny_score(xt, t).detach().requires_grad_(True)
    out = ddim.ddim_step(xt, et, at, at_next)
    gout = T(g['w'])
    ga, gb = torch.autograd.grad(out, (xt, et), gout)
    ma, mb = ddim.ddim_step_vjp(gout, xt.detach(), et.detach(), at, at_next)
    assert torch.equal(ga, ma) and torch.equal(gb, mb)


def _hmc_case(golden, deg):
    g = golden(f'g4_hmc_{deg}_32.npz')
    if deg == 'inpaint':
        op = operators.InpaintRef(3, 32, T(g['missing']))
    elif deg == 'sr4':
        op = operators.BlockMeanRef(3, 32, 4)
    else:
        op = operators.SpectralBlurRef(*(T(g[f'aniso_{k}']) for k in ('U1', 'U2', 'V1', 'V2', 'D')))
    return g, op


def test_g4_full_hmc_inpaint_bit_exact(golden, tiny_score):
    """Whole reference run (208 trajectories, 100 accepts) reproduced bit for bit."""
    g, op = _hmc_case(golden, 'inpaint')
    torch.manual_seed(int(g['seed']))
    trace = {}
    out = hmc_ref.hmc_reference(T(g['x']), schedule.betas_fp32(), SEQ, SEQ_NEXT, tiny_score, op,
                                T(g['y_0']), T(g['x_orig']), tau=float(g['tau']), epsilon=float(g['epsilon']),
                                m=float(g['m']), sigma_0=float(g['sigma_0']), trace=trace)
    assert np.array_equal(out.numpy(), g['out'])
    assert np.array_equal(-np.array(trace['dH'], dtype=np.float32), g['neg_dH'].astype(np.float32))
    assert len(trace['accept']) == len(g['u']) and sum(trace['accept']) == 100
    assert np.allclose(trace['psnr'], g['psnr'], rtol=0, atol=1e-5)


@pytest.mark.parametrize('deg', ['inpaint', 'sr4', 'aniso'])
def test_g4_first_trajectory(golden, tiny_score, deg):
    """Per-chain `trajectory` against the reference's first outer iteration."""
    g, op = _hmc_case(golden, deg)
    sig0 = float(g['sigma_0'])
    out = hmc_ref.trajectory(T(g['x']), T(g['p0']), schedule.betas_fp32(), SEQ, SEQ_NEXT, tiny_score, op,
                             T(g['y_0']), sigma_y=hmc_ref.sigma_y_at(0, sig0), eps=0.05, m=1.0, L=20)
    tol = 0.0 if deg == 'inpaint' else 2e-5
    assert rel(out['x'].numpy(), g['pos_last']) <= tol
    assert rel(out['xt'].numpy(), g['dec_last']) <= tol
    dH = float((out['H1'] - out['H0'])[0])
    assert abs(dH - (-g['neg_dH'][0])) <= (0.0 if deg == 'inpaint' else 0.05)


def test_g4_hmc_chains_b1_takes_reference_decisions(golden, tiny_score):
    """The per-chain generalisation, fed the reference's own noise, makes the reference's decisions."""
    g, op = _hmc_case(golden, 'inpaint')
    torch.manual_seed(int(g['seed']))
    us = g['u']

    def draw_p(it):
        p = torch.randn(1, 3, 32, 32)
        torch.rand(1)                      # keep the CPU generator in step with the reference's draw order
        return p

    trace = {}
    out = hmc_ref.hmc_chains(T(g['x']), schedule.betas_fp32(), SEQ, SEQ_NEXT, tiny_score, op, T(g['y_0']),
                             T(g['x_orig']), tau=1.0, epsilon=0.05, m=1.0, sigma_0=float(g['sigma_0']),
                             draw_p=draw_p, draw_u=lambda it: torch.tensor([us[it]], dtype=torch.float32),
                             trace=trace)
    assert np.array_equal(out[0].numpy(), g['out'])


def test_g5_operators_256(golden):
    g = golden('g5_ops_256.npz')
    x = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(int(g['x_seed']))) * 2 - 1
    ops = dict(inpaint=operators.InpaintRef(3, 256, T(g['missing']).long()),
               sr4=operators.BlockMeanRef(3, 256, 4), sr16=operators.BlockMeanRef(3, 256, 16))
    assert ops['inpaint'].M == 15729
    for name, op in ops.items():
        hx = op.H(x)
        y = torch.randn(hx.shape, generator=torch.Generator().manual_seed(99))
        hty = op.Ht(y)
        tol = 0.0 if name == 'inpaint' else 2e-6
        assert rel(hx[0, T(g[f'{name}_probe_m'])].numpy(), g[f'{name}_Hx_probe']) <= tol
        assert rel(hty[0, T(g[f'{name}_probe_n'])].numpy(), g[f'{name}_Hty_probe']) <= tol
        assert abs(float(hx.double().norm()) / float(g[f'{name}_Hx_norm']) - 1) < 1e-6


def test_g14_oracle_reproduces_the_reference_run_with_the_f64_score(golden, tiny_score):
    """G14 = the reference's `hmc()` with the tiny score evaluated in float64 (the tape the GPU whole-run test replays):
    the oracle reproduces it bit for bit too."""
    from oracle.tiny_score import F64Score
    g = golden('g14_hmc_f64_sr4_32.npz')
    torch.manual_seed(int(g['seed']))
    trace = {}
    out = hmc_ref.hmc_reference(T(g['x']), schedule.betas_fp32(), SEQ, SEQ_NEXT, F64Score(tiny_score), operators.BlockMeanRef(3, 32, 4),
                                T(g['y_0']), T(g['x_orig']), tau=float(g['tau']), epsilon=float(g['epsilon']),
                                m=float(g['m']), sigma_0=float(g['sigma_0']), trace=trace)
    assert len(trace['accept']) == len(g['u']) and sum(trace['accept']) == 100
    assert np.allclose(-np.array(trace['dH']), g['neg_dH'], rtol=0, atol=0.05)
    assert rel(out.numpy(), g['out']) <= 2e-5


def _g15_oracle_op(g, deg, dim):
    if deg == 'color':
        return operators.ColorRef(dim)
    if deg == 'gauss':
        return operators.SpectralBlurRef(T(g['gauss_U']), T(g['gauss_U']), T(g['gauss_V']), T(g['gauss_V']), T(g['gauss_D']))
    if deg == 'cs4':
        return operators.WalshHadamardRef(3, dim, int(g['ratio']), T(g['perm']))
    if deg == 'box':
        return operators.InpaintRef(3, dim, T(g['box_missing']))
    if deg == 'sr16':
        return operators.BlockMeanRef(3, dim, 16)
    return operators.SeparableStridedRef(T(g['kernel']), 3, dim, int(g['factor']))


@pytest.mark.parametrize('deg,dim', [('color', 32), ('cs4', 32), ('sr16', 32)])
def test_g15_oracle_reproduces_the_reference_runs_of_the_remaining_operators(golden, tiny_score, deg, dim):
    """G15 = the reference's whole `hmc()` run per remaining operator (oracle/gen_golden_hmc_ops2.py).  Three of the six
    on the CPU (the suite's time budget); all six are replayed on the GPU (tests/test_reference_run_gpu.py)."""
    from oracle.tiny_score import F64Score
    g = golden(f'g15_hmc_f64_{deg}_{dim}.npz')
    torch.manual_seed(int(g['seed']))
    trace = {}
    out = hmc_ref.hmc_reference(T(g['x']), schedule.betas_fp32(), SEQ, SEQ_NEXT, F64Score(tiny_score), _g15_oracle_op(g, deg, dim),
                                T(g['y_0']), T(g['x_orig']), tau=float(g['tau']), epsilon=float(g['epsilon']),
                                m=float(g['m']), sigma_0=float(g['sigma_0']), trace=trace)
    assert len(trace['accept']) == len(g['u']) and sum(trace['accept']) == 100
    assert np.allclose(-np.array(trace['dH']), g['neg_dH'], rtol=0, atol=0.05)
    assert rel(out.numpy(), g['out']) <= 2e-5
