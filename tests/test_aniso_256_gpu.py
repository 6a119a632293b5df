"""GPU: the MFMA spectral chain at BASELINE configs[3] size against outputs captured from the reference object (G13),
and the reference's 256 x 256 probes of the inpainting / SR operators (G5) through the HIP kernels."""
import numpy as np
import pytest
import torch

from oracle import operators as oops
from tests.test_aniso_256_cpu import _rel, _xy

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _check(op, g):
    x, y = _xy(g)
    pm = T(g['probe']).long()
    hx, hty, hpy = op.H(x.cuda()).cpu(), op.Ht(y.cuda()).cpu(), op.H_pinv(y.cuda()).cpu()
    assert _rel(hx[:, pm], T(g['Hx_probe'])) < 2e-5 and _rel(hty[:, pm], T(g['Hty_probe'])) < 2e-5
    assert _rel(hpy[:, pm], T(g['Hpinv_probe'])) < 1e-4
    assert np.allclose(hx.double().norm(dim=1).numpy(), g['Hx_norm'], rtol=1e-5)
    assert np.allclose(hty.double().norm(dim=1).numpy(), g['Hty_norm'], rtol=1e-5)
    # data term on the same operator: loss = |y - H clip(x)|^2, gradient = -2 H^T r (x) mask, against the probes' operator
    loss, grad = op.data_term((1.5 * x).cuda().contiguous(), y.cuda().contiguous(), apply_clip=True)
    r = y - op.H((1.5 * x).clip(-1, 1).cuda()).cpu()
    assert _rel(loss.cpu(), (r.double() ** 2).sum(1)) < 1e-5
    want = (-2 * op.Ht(r.cuda()).cpu()).reshape(x.shape) * ((1.5 * x).abs() <= 1)
    assert _rel(grad.cpu(), want) < 2e-5


def test_exported_reference_operator_on_the_mfma_chain(golden):
    from nhmc import operators
    g = golden('g13_aniso_256.npz')
    D = oops.SpectralBlurRef.multiplier_map(T(g['s_sorted']), T(g['perm'].astype(np.int64)), 3, 256)
    _check(operators.Deblurring2D.from_factors(T(g['U1']), T(g['U2']), T(g['V1']), T(g['V2']), D, 'cuda'), g)


def test_production_constructor_on_this_host(golden):
    """`build_operator('deblur_aniso')` issues the reference's svd / sort calls on the host.  On a host whose LAPACK /
    sort give the fixture's factors it IS the reference operator (bit-identical in the build container,
    tests/test_aniso_256_cpu.py); elsewhere the singular vectors of the near-null space differ and so does this
    misaligned operator -- as the reference's own instances do between machines.  Checked here: whichever holds."""
    from nhmc import operators
    g = golden('g13_aniso_256.npz')
    op = operators.build_operator('deblur_aniso', 3, 256, 'cuda')
    same = all(np.array_equal(op.factors[i].cpu().numpy(), g[k]) for i, k in enumerate(('U1', 'U2', 'V1', 'V2')))
    if same:
        _check(op, g)
    else:                                      # another host: a self-consistent instance of the same construction
        x, y = _xy(g)
        lhs = (op.H(x.cuda()).double() * y.cuda().double()).sum()
        rhs = (x.cuda().reshape(2, -1).double() * op.Ht(y.cuda()).double()).sum()
        assert abs(float(lhs - rhs)) < 1e-3 * (1 + abs(float(lhs)))
        assert int((op.Dmap != 0).sum()) == 148458
    print('production deblur_aniso equals the reference CPU instance on this host:', same)


def test_g5_probes_through_the_kernels(golden):
    """The reference's own H / Ht outputs at 256 x 256 for inpaint_random, sr4, sr16 (G5) on the GPU."""
    from nhmc import operators
    g = golden('g5_ops_256.npz')
    x = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(int(g['x_seed']))) * 2 - 1
    ops = dict(inpaint=operators.Inpainting(3, 256, T(g['missing']).long(), 'cuda'),
               sr4=operators.SuperResolution(3, 256, 4, 'cuda'), sr16=operators.SuperResolution(3, 256, 16, 'cuda'))
    for name, op in ops.items():
        hx = op.H(x.cuda()).cpu()
        y = torch.randn(hx.shape, generator=torch.Generator().manual_seed(99))
        hty = op.Ht(y.cuda()).cpu()
        tol = 0.0 if name == 'inpaint' else 2e-6
        assert _rel(hx[0, T(g[f'{name}_probe_m']).long()], T(g[f'{name}_Hx_probe'])) <= tol
        assert _rel(hty[0, T(g[f'{name}_probe_n']).long()], T(g[f'{name}_Hty_probe'])) <= tol
        assert abs(float(hx.double().norm()) / float(g[f'{name}_Hx_norm']) - 1) < 1e-6
    # and the aniso probes of G5 (same reference construction as G13, different input)
    ga = golden('g13_aniso_256.npz')
    D = oops.SpectralBlurRef.multiplier_map(T(ga['s_sorted']), T(ga['perm'].astype(np.int64)), 3, 256)
    an = operators.Deblurring2D.from_factors(T(ga['U1']), T(ga['U2']), T(ga['V1']), T(ga['V2']), D, 'cuda')
    hx = an.H(x.cuda()).cpu()
    y = torch.randn(hx.shape, generator=torch.Generator().manual_seed(99))
    hty = an.Ht(y.cuda()).cpu()
    assert _rel(hx[0, T(g['aniso_probe_m']).long()], T(g['aniso_Hx_probe'])) < 2e-5
    assert _rel(hty[0, T(g['aniso_probe_n']).long()], T(g['aniso_Hty_probe'])) < 2e-5


def test_two_products_per_launch_equal_the_one_product_chain(golden):
    """k_pair256 (the intermediate of each product pair stays in LDS) against the one-product-per-launch chain
    (NHMC_SPECTRAL_PAIRS=0): same MFMA instruction, same k order -> the same bits for H, the data term and the form fused
    with the last DDIM-step VJP."""
    import os
    import nhmc.kernels as K
    from nhmc import operators
    g = golden('g13_aniso_256.npz')
    D = oops.SpectralBlurRef.multiplier_map(T(g['s_sorted']), T(g['perm'].astype(np.int64)), 3, 256)
    op = operators.Deblurring2D.from_factors(T(g['U1']), T(g['U2']), T(g['V1']), T(g['V2']), D, 'cuda')
    gen = torch.Generator().manual_seed(77)
    B = 3
    x = (torch.randn(B, 3, 256, 256, generator=gen) * 0.7).cuda()
    e = torch.randn(B, 6, 256, 256, generator=gen).cuda()
    y = torch.randn(B, 3 * 256 * 256, generator=gen).cuda()
    at, atn = torch.full((B,), 0.5214230418).cuda(), torch.ones(B).cuda()
    cur = K.ddim_mix_fwd(x, e, at, atn, final_clip=True)['xt_next']

    def run():
        ge = torch.zeros_like(e)
        h = op.H(x)
        loss, gr = op.data_term(x, y, apply_clip=True)
        loss2, gx, _ = op.fused_last_vjp(x, e, at, atn, y, g_e_out=ge, xt_next=cur)
        return h, loss, gr, loss2, gx, ge
    old = os.environ.get('NHMC_SPECTRAL_PAIRS')
    try:
        os.environ['NHMC_SPECTRAL_PAIRS'] = '1'
        a = run()
        os.environ['NHMC_SPECTRAL_PAIRS'] = '0'
        b = run()
    finally:
        os.environ.pop('NHMC_SPECTRAL_PAIRS', None)
        if old is not None:
            os.environ['NHMC_SPECTRAL_PAIRS'] = old
    for u, v in zip(a, b):
        assert torch.equal(u, v) or _rel(u.cpu(), v.cpu()) < 1e-6, _rel(u.cpu(), v.cpu())
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and torch.equal(a[4], b[4])
