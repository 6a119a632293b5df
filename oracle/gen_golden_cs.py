"""Generate tests/golden/g9_cs_{32,64}.npz by running the reference's WalshHadamardCS operator (build container only;
same rules as oracle/gen_golden.py)."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.gen_golden import import_reference, np32, save  # noqa: E402


def main():
    import_reference()
    from obs_functions.Hfuncs import WalshHadamardCS
    for dim, ratio in ((32, 4), (64, 2)):
        g = torch.Generator().manual_seed(900 + dim)
        perm = torch.randperm(dim * dim, generator=g)
        op = WalshHadamardCS(3, dim, ratio, perm, 'cpu')
        x = torch.randn(2, 3, dim, dim, generator=g)
        hx = op.H(x)
        y = torch.randn(hx.shape, generator=g)
        save(f'g9_cs_{dim}.npz', x=np32(x), perm=np32(perm), ratio=ratio, Hx=np32(hx), y=np32(y),
             Hty=np32(op.Ht(y.clone())), Hpinvy=np32(op.H_pinv(y.clone())))


if __name__ == '__main__':
    main()
