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
    errs = [rel(op.H(T(g['x']).cuda()), T(g['Hx'])), rel(op.Ht(T(g['y']).cuda()), T(g['Hty'])), rel(op.H_pinv(T(g['y']).cuda()), T(g['Hpinvy']))]
    print(f'SRConv {dim}: H / Ht / H_pinv against the reference outputs: {errs}')
    # the SVD is recomputed on this host (LAPACK may differ from the generating host's in the last bits); with the same
    # factors the MFMA chain is the reference's arithmetic (test below)
    assert max(errs) < 2e-5


def test_srconv_with_the_reference_factors_is_bit_identical(golden):
    """G15's bicubic fixture carries the reference instance's own SVD factors: with them H / Ht / H_pinv and the data
    term's gradient are the CPU restatement's bits (staged products in the reference's order, right factor first in the
    adjoint as autograd has it; every MFMA product an exact k-ascending FMA chain like torch's CPU sgemm)."""
    from nhmc import operators
    g = golden('g15_hmc_f64_bicubic2_64.npz')
    dim, f = 64, int(g['factor'])
    svd = (T(g['srconv_U']), T(g['srconv_s']), T(g['srconv_V']))
    op = operators.SRConv.from_svd(*svd, 3, dim, 'cuda', stride=f)
    ref = oops.SeparableStridedRef(T(g['kernel']), 3, dim, f, svd=svd)
    g_ = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, dim, dim, generator=g_) * 0.8
    y = torch.randn(2, ref.M, generator=g_)
    assert torch.equal(op.H(x.cuda()).cpu(), ref.H(x))
    assert torch.equal(op.Ht(y.cuda()).cpu(), ref.Ht(y))
    assert torch.equal(op.H_pinv(y.cuda()).cpu(), ref.H_pinv(y))
    loss_ref, g_ref = hmc_ref.data_term(x, ref, y)
    loss, gx = op.data_term(x.cuda(), y.cuda(), apply_clip=True)
    assert torch.equal(gx.cpu(), g_ref), float((gx.cpu() - g_ref).abs().max() / g_ref.abs().max())
    assert rel(loss, loss_ref) < 1e-6


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
