"""GPU: Colorization and Deblurring (deblur_gauss) on the HIP kernels against the reference's outputs (G8) and
the oracle's autograd data term."""
import pytest
import torch

from oracle import hmc_ref, operators as oops

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('dim', [32, 64])
def test_color_and_gauss_against_reference_outputs(golden, dim):
    from nhmc import operators
    g = golden(f'g8_ops2_{dim}.npz')
    x = T(g['x']).cuda()
    gauss = operators.Deblurring2D.from_factors(T(g['gauss_U']), T(g['gauss_U']), T(g['gauss_V']), T(g['gauss_V']),
                                                T(g['gauss_D']), 'cuda')
    for name, op, tol in (('color', operators.Colorization(dim, 'cuda'), 0.0), ('gauss', gauss, 2e-5)):
        y = T(g[f'{name}_y']).cuda()
        # colorization: V^T, s and U applied in the reference's rounding order -> the reference's bits (tol 0)
        assert rel(op.H(x), T(g[f'{name}_Hx'])) <= tol, name
        assert rel(op.Ht(y), T(g[f'{name}_Hty'])) <= tol, name
        assert rel(op.H_pinv(y), T(g[f'{name}_Hpinvy'])) <= 5 * tol, name


@pytest.mark.parametrize('dim,B', [(32, 3), (256, 2)])
def test_color_data_term(dim, B):
    from nhmc import operators
    ref, op = oops.ColorRef(dim), operators.Colorization(dim, 'cuda')
    g_ = torch.Generator().manual_seed(5)
    xt = torch.randn(B, 3, dim, dim, generator=g_) * 0.8
    y = torch.randn(B, ref.M, generator=g_)
    loss_ref, g_ref = hmc_ref.data_term(xt, ref, y)
    loss, g = op.data_term(xt.cuda(), y.cuda(), apply_clip=True)
    assert rel(loss, loss_ref) < 2e-6 and torch.equal(g.cpu(), g_ref)          # the gradient in autograd's rounding order


def test_deblur_gauss_constructor_builds_the_reference_operator(golden):
    """Own construction from the 5-tap kernel: same band matrix / truncation / multiplier multiset as the reference
    object, and a self-consistent operator (adjointness)."""
    from nhmc import operators
    g = golden('g8_ops2_64.npz')
    op = operators.build_operator('deblur_gauss', 3, 64, 'cuda')
    for c in range(3):
        a = torch.sort(op.Dmap[c].reshape(-1).cpu()).values
        b = torch.sort(T(g['gauss_D'][c]).reshape(-1)).values
        assert float((a - b).abs().max()) < 1e-5
    x = torch.randn(2, 3, 64, 64, device='cuda')
    y = torch.randn(2, 3 * 64 * 64, device='cuda')
    lhs = (op.H(x).double() * y.double()).sum()
    rhs = (x.reshape(2, -1).double() * op.Ht(y).double()).sum()
    assert abs(float(lhs - rhs)) < 1e-3 * (1 + abs(float(lhs)))
