"""GPU: Walsh-Hadamard CS operator (FWHT kernels) against the reference's outputs (G9) and the oracle."""
import pytest
import torch

from oracle import hmc_ref, operators as oops

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('dim', [32, 64])
def test_cs_against_reference_outputs_bit_exact(golden, dim):
    from nhmc import operators
    g = golden(f'g9_cs_{dim}.npz')
    op = operators.WalshHadamardCS(3, dim, int(g['ratio']), T(g['perm']), 'cuda')
    assert torch.equal(op.H(T(g['x']).cuda()).cpu(), T(g['Hx']))
    assert torch.equal(op.Ht(T(g['y']).cuda()).cpu(), T(g['Hty']))
    assert torch.equal(op.H_pinv(T(g['y']).cuda()).cpu(), T(g['Hpinvy']))


@pytest.mark.parametrize('dim,ratio,B', [(16, 2, 2), (32, 4, 3), (128, 4, 2), (256, 4, 2), (256, 16, 1)])
def test_cs_data_term(dim, ratio, B):
    from nhmc import operators
    g_ = torch.Generator().manual_seed(dim + ratio)
    perm = torch.randperm(dim * dim, generator=g_)
    ref, op = oops.WalshHadamardRef(3, dim, ratio, perm), operators.WalshHadamardCS(3, dim, ratio, perm, 'cuda')
    xt = torch.randn(B, 3, dim, dim, generator=g_) * 0.8
    y = torch.randn(B, ref.M, generator=g_)
    assert torch.equal(op.H(xt.cuda()).cpu(), ref.H(xt))
    assert torch.equal(op.Ht(y.cuda()).cpu(), ref.Ht(y))
    loss_ref, g_ref = hmc_ref.data_term(xt, ref, y)
    loss, g = op.data_term(xt.cuda(), y.cuda(), apply_clip=True)
    # the adjoint runs its butterfly stages in autograd's (descending) order: the oracle's gradient, bit for bit
    assert rel(loss, loss_ref) < 2e-6 and torch.equal(g.cpu(), g_ref)
