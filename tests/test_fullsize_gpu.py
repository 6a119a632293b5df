"""GPU: size-independent properties at BASELINE's full per-GPU sizes (64 chains of 3x256x256) for the sr4
(configs[2]) and deblur_aniso (configs[3]) data terms: adjointness <Hx, y> = <x, H^T y>, pseudo-inverse
identities, and the fused data term = -2 H^T (y - H clip x) (x) mask with loss = |r|^2."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('deg,tol', [('sr4', 1e-6), ('sr16', 1e-6), ('deblur_aniso', 5e-5), ('cs4', 1e-6), ('color', 1e-6)])
def test_full_size_operator_properties_b64(deg, tol):
    import nhmc.kernels as K
    from nhmc import operators
    B, shape = 64, (64, 3, 256, 256)
    dev = torch.device('cuda')
    op = operators.build_operator(deg, 3, 256, dev, generator=torch.Generator().manual_seed(1))
    x = K.randn_philox(shape, 11, 0, 0).mul_(0.8)
    y = torch.randn(B, op.M, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    Hx, Hty = op.H(x), op.Ht(y)
    lhs = (Hx.double() * y.double()).sum(1)
    rhs = (x.reshape(B, -1).double() * Hty.double()).sum(1)
    assert float(((lhs - rhs).abs() / (lhs.abs() + 1)).max()) < 50 * tol                      # adjoint pair, per chain
    if deg.startswith('sr') or deg.startswith('cs') or deg == 'color':
        assert rel(op.H(op.H_pinv(y).reshape(shape)), y) < 20 * tol                              # H H^+ = I (full row rank)
    loss, g = op.data_term(x, y, apply_clip=True)
    r = y - op.H(x.clip(-1, 1))
    assert rel(loss, (r.double() ** 2).sum(1)) < 20 * tol
    mask = ((x >= -1) & (x <= 1)).float().reshape(B, -1)
    want = -(2 * op.Ht(r)) * mask
    assert rel(g.reshape(B, -1), want) < 50 * tol
    # chains are independent: a permutation of the batch permutes the outputs
    perm = torch.randperm(B, device=dev)
    loss_p, g_p = op.data_term(x[perm].contiguous(), y[perm].contiguous(), apply_clip=True)
    assert torch.equal(loss_p, loss[perm]) and torch.equal(g_p, g[perm])
