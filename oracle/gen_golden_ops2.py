"""Generate tests/golden/g8_ops2_{32,64}.npz by running the reference's Colorization and Deblurring (deblur_gauss)
operators (build container only; same rules as oracle/gen_golden.py)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.gen_golden import import_reference, np32, save  # noqa: E402


def main():
    import_reference()
    from obs_functions.Hfuncs import Colorization, Deblurring
    for dim in (32, 64):
        g = torch.Generator().manual_seed(800 + dim)
        x = torch.randn(2, 3, dim, dim, generator=g)
        sigma = 10
        pdf = lambda v: torch.exp(torch.Tensor([-0.5 * (v / sigma) ** 2]))
        kernel = torch.Tensor([pdf(-2), pdf(-1), pdf(0), pdf(1), pdf(2)])
        kernel = kernel / kernel.sum()                                  # main_sampling.py:308-314
        ops = dict(color=Colorization(dim, 'cpu'), gauss=Deblurring(kernel, 3, dim, 'cpu'))
        arrays = dict(x=np32(x), kernel=np32(kernel))
        for name, op in ops.items():
            hx = op.H(x)
            y = torch.randn(hx.shape, generator=g)
            arrays[f'{name}_Hx'], arrays[f'{name}_y'] = np32(hx), np32(y)
            arrays[f'{name}_Hty'], arrays[f'{name}_Hpinvy'] = np32(op.Ht(y.clone())), np32(op.H_pinv(y.clone()))
        db = ops['gauss']
        hw = dim * dim
        sing = db.singulars()
        D = torch.zeros(3, hw)
        for c in range(3):
            D[c, db._perm] = sing[3 * torch.arange(hw) + c]
        arrays.update(gauss_U=np32(db.U_small), gauss_V=np32(db.V_small), gauss_D=np32(D.reshape(3, dim, dim)))
        save(f'g8_ops2_{dim}.npz', **arrays)


if __name__ == '__main__':
    main()
