"""Generate tests/golden/g10_srconv_{64,128}.npz by running the reference's SRConv (sr_bicubic4) operator
(build container only; same rules as oracle/gen_golden.py)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.gen_golden import import_reference, np32, save  # noqa: E402


def bicubic(factor):
    def bk(x, a=-0.5):                                               # main_sampling.py:266-272
        if abs(x) <= 1:
            return (a + 2) * abs(x) ** 3 - (a + 3) * abs(x) ** 2 + 1
        elif 1 < abs(x) and abs(x) < 2:
            return a * abs(x) ** 3 - 5 * a * abs(x) ** 2 + 8 * a * abs(x) - 4 * a
        return 0
    k = np.zeros((factor * 4))
    for i in range(factor * 4):
        k[i] = bk((1 / factor) * (i - np.floor(factor * 4 / 2) + 0.5))
    k = k / np.sum(k)
    kernel = torch.from_numpy(k).float()
    return kernel / kernel.sum()


def main():
    import_reference()
    from obs_functions.Hfuncs import SRConv
    for dim, factor in ((128, 4), (64, 2)):
        kernel = bicubic(factor)
        op = SRConv(kernel, 3, dim, 'cpu', stride=factor)
        g = torch.Generator().manual_seed(1000 + dim)
        x = torch.randn(2, 3, dim, dim, generator=g)
        hx = op.H(x)
        y = torch.randn(hx.shape, generator=g)
        save(f'g10_srconv_{dim}.npz', x=np32(x), kernel=np32(kernel), factor=factor, Hx=np32(hx), y=np32(y),
             Hty=np32(op.Ht(y.clone())), Hpinvy=np32(op.H_pinv(y.clone())))


if __name__ == '__main__':
    main()
