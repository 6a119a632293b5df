"""GPU: SRConv (sr_bicubic) on the rectangular MFMA chain against the reference's outputs (G10) and the oracle."""
import pytest
import torch

from oracle import hmc_ref, operators as oops

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('dim', [64, 128])
def test_srconv_against_reference_outputs(golden, dim):
    from nhmc import operators
    g = golden(f'g10_srconv_{dim}.npz')
    op = operators.SRConv(T(g['kernel']), 3, dim, 'cuda', stride=int(g['factor']))
    assert rel(op.H(T(g['x']).cuda()), T(g['Hx'])) < 2e-5
    assert rel(op.Ht(T(g['y']).cuda()), T(g['Hty'])) < 2e-5
    assert rel(op.H_pinv(T(g['y']).cuda()), T(g['Hpinvy'])) < 2e-5


@pytest.mark.parametrize('dim,factor,B', [(64, 2, 2), (128, 4, 3), (256, 4, 2)])
def test_srconv_data_term(dim, factor, B):
    from nhmc import operators
    k = operators.bicubic_taps(factor)
    ref, op = oops.SeparableStridedRef(k, 3, dim, factor), operators.build_operator(f'sr_bicubic{factor}', 3, dim, 'cuda')
    g_ = torch.Generator().manual_seed(dim)
    xt = torch.randn(B, 3, dim, dim, generator=g_) * 0.8
    y = torch.randn(B, ref.M, generator=g_)
    loss_ref, g_ref = hmc_ref.data_term(xt, ref, y)
    loss, g = op.data_term(xt.cuda(), y.cuda(), apply_clip=True)
    assert rel(loss, loss_ref) < 2e-5 and rel(g, g_ref) < 2e-5
