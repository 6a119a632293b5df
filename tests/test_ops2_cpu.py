"""CPU: oracle restatements of Colorization and Deblurring (deblur_gauss) against the reference's outputs (G8)."""
import numpy as np
import pytest
import torch

from oracle import operators as oops

T = torch.from_numpy


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(1e-30, np.abs(b).max())


@pytest.mark.parametrize('dim', [32, 64])
def test_g8_color_and_gauss(golden, dim):
    g = golden(f'g8_ops2_{dim}.npz')
    x = T(g['x'])
    ops = dict(color=oops.ColorRef(dim),
               gauss=oops.SpectralBlurRef(T(g['gauss_U']), T(g['gauss_U']), T(g['gauss_V']), T(g['gauss_V']), T(g['gauss_D'])))
    for name, op in ops.items():
        y = T(g[f'{name}_y'])
        assert rel(op.H(x).numpy(), g[f'{name}_Hx']) < 2e-6, name
        assert rel(op.Ht(y).numpy(), g[f'{name}_Hty']) < 2e-6, name
        assert rel(op.H_pinv(y).numpy(), g[f'{name}_Hpinvy']) < 5e-6, name
    # the gauss band matrix uses 4 of its 5 taps (half-open range, Hfuncs.py:250)
    assert int((oops.band_matrix(T(g['kernel']), dim)[dim // 2] != 0).sum()) == 4
