"""CPU: oracle restatement of SRConv (sr_bicubic) against the reference's outputs (G10)."""
import numpy as np
import pytest
import torch

from oracle import operators as oops

T = torch.from_numpy


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(1e-30, np.abs(b).max())


@pytest.mark.parametrize('dim', [64, 128])
def test_g10_srconv(golden, dim):
    g = golden(f'g10_srconv_{dim}.npz')
    op = oops.SeparableStridedRef(T(g['kernel']), 3, dim, int(g['factor']))
    # the reference's stage order: bit-identical on the torch build that wrote G10; 1e-6 leaves room for another
    # host's BLAS blocking (round 2's collapsed form needed 2e-4 for H^+)
    assert rel(op.H(T(g['x'])).numpy(), g['Hx']) < 1e-6
    assert rel(op.Ht(T(g['y'])).numpy(), g['Hty']) < 1e-6
    assert rel(op.H_pinv(T(g['y'])).numpy(), g['Hpinvy']) < 1e-6


def test_product_bicubic_taps_match_reference_kernel(golden):
    from nhmc import operators
    g = golden('g10_srconv_128.npz')
    assert np.allclose(operators.bicubic_taps(4).numpy(), g['kernel'], atol=1e-7)
    Hs = operators.strided_conv_matrix(T(g['kernel']), 128, 4)
    assert Hs.shape == (32, 128) and abs(float(Hs.sum(1).mean()) - 1.0) < 1e-5
