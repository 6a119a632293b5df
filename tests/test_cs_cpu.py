"""CPU: oracle restatement of the Walsh-Hadamard CS operator against the reference's outputs (G9), bit for bit."""
import numpy as np
import pytest
import torch

from oracle import operators as oops

T = torch.from_numpy


@pytest.mark.parametrize('dim', [32, 64])
def test_g9_walsh_hadamard(golden, dim):
    g = golden(f'g9_cs_{dim}.npz')
    op = oops.WalshHadamardRef(3, dim, int(g['ratio']), T(g['perm']))
    assert np.array_equal(op.H(T(g['x'])).numpy(), g['Hx'])
    assert np.array_equal(op.Ht(T(g['y'])).numpy(), g['Hty'])
    assert np.array_equal(op.H_pinv(T(g['y'])).numpy(), g['Hpinvy'])
    # orthonormal rows: H H^T = I
    y = T(g['y'])
    assert torch.allclose(op.H(op.Ht(y).reshape(2, 3, dim, dim)), y, atol=1e-5)
